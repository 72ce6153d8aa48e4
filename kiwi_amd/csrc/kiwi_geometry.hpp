// kiwi_geometry.hpp -- geometry_kernel / cellgroup_kernel: one thread per (source, receiver, centroid): spherical
// azimuth / distance update (orthodrome.f90:77-156), moment-tensor weights (seismogram.f90:316-336), GF grid indices and
// bilinear weights (gfdb.f90:781-815), time shift split (sparse_trace.f90:640-645); writes one 80-byte GeoRec and the load
// descriptors / interpolation coefficients the accumulate kernels read.  ALWAYS built with -ffp-contract=off, whatever the
// arithmetic mode of the accumulate kernels: the integer fields (rows, shifts) and the weights are the reference's bit for bit.
#pragma once
#include "kiwi_common.hpp"
#include "kiwi_libm32.hpp"

namespace kiwi {

// ------------------------------------------------------------------------------------------------
// geometry

__device__ __forceinline__ double clipd(double x, double mi, double ma) { return fmin(fmax(mi, x), ma); }
__device__ __forceinline__ double wrapd(double x, double mi, double ma) { return x - floor((x - mi) / (ma - mi)) * (ma - mi); }

// The default-real libm calls of the reference host (sin, cos, atan2 -> glibc sinf, cosf, atan2f)
// are reproduced bit for bit by kiwi_libm32.hpp; the real*8 ones (sin, cos, acos, asin) use the
// device's fp64 libm, which agrees with glibc to an ulp of fp64 ("tolerances").
__device__ __forceinline__ float sin32(float x) { return libm32::sinf_glibc(x); }
__device__ __forceinline__ float cos32(float x) { return libm32::cosf_glibc(x); }


// Length of the centroid group that accumulate_grouped_kernel forms when a group STARTS at centroid c of a source: the
// following centroids at this same point whose integer shifts stay within the LDS halo (also returns the shift range).
__device__ __forceinline__ int group_len(const float *__restrict__ cent, int c0, int nc, int c, float dt, int &smin, int &smax)
{
    const float *ce = cent + (size_t)(c0 + c) * 10;
    const float dnorth = ce[0], deast = ce[1], depth = ce[2];
    int len = 1;
    smin = smax = (int)floorf(ce[3] / dt);
    for (int k = c + 1; k < nc && len < kMaxGroup; k++) {
        const float *ne = cent + (size_t)(c0 + k) * 10;
        if (!(ne[0] == dnorth && ne[1] == deast && ne[2] == depth)) break;
        const int sh = (int)floorf(ne[3] / dt);
        const int nmin = min(smin, sh), nmax = max(smax, sh);
        if (nmax - nmin > kHalo - 10) break;
        smin = nmin; smax = nmax; len++;
    }
    return len;
}

// Does the grouped kernel start a group at centroid c?  Groups never span two points, so the first centroid of a
// same-point run always starts one; inside a run the starts follow from the group lengths (all centroids of a run share
// their GF rows, so they are stored or missing together and the kernel's skipping of missing ones does not interfere).
__device__ __forceinline__ bool starts_group(const float *__restrict__ cent, int c0, int nc, int c, float dt)
{
    const float *ce = cent + (size_t)(c0 + c) * 10;
    int r0 = c;
    while (r0 > 0) {
        const float *pe = cent + (size_t)(c0 + r0 - 1) * 10;
        if (!(pe[0] == ce[0] && pe[1] == ce[1] && pe[2] == ce[2])) break;
        r0--;
    }
    int pos = r0, lo, hi;
    while (pos < c) pos += group_len(cent, c0, nc, pos, dt, lo, hi);
    return pos == c;
}

// A centroid's interpolation-coefficient line (kCoefLine floats, see kiwi_common.hpp): wl = (1 - w) factor, wr = w factor per GF
// component in application order, each rounded on its own (sparse_trace.f90:643-647 with the factors of seismogram.f90:171-250)
template <int NG>
__device__ __forceinline__ void coef_line(const GeoRec &g, float sd, float (&cf)[20])
{
    {
        const float wr0 = g.wfrac, wl0 = 1.f - g.wfrac;
        const float fd[4] = { g.f[0] * sd, g.f[1] * sd, g.f[2] * sd, g.f[5] * sd };
        int i = 0;
        if (NG == 10) {
            const float fh[6] = { g.f[0], g.f[1], g.f[2], g.f[5], g.f[3], g.f[4] };
#pragma unroll
            for (int q = 0; q < 6; q++, i++) { cf[coef_wl<10>(i)] = wl0 * fh[q]; cf[coef_wr<10>(i)] = wr0 * fh[q]; }
#pragma unroll
            for (int q = 0; q < 4; q++, i++) { cf[coef_wl<10>(i)] = wl0 * fd[q]; cf[coef_wr<10>(i)] = wr0 * fd[q]; }
        } else {
            const float fh8[5] = { g.f[0], g.f[1], g.f[2], g.f[3], g.f[4] };
#pragma unroll
            for (int q = 0; q < 5; q++, i++) { cf[coef_wl<8>(i)] = wl0 * fh8[q]; cf[coef_wr<8>(i)] = wr0 * fh8[q]; }
#pragma unroll
            for (int q = 0; q < 3; q++, i++) { cf[coef_wl<8>(i)] = wl0 * fd[q]; cf[coef_wr<8>(i)] = wr0 * fd[q]; }
#pragma unroll
            for (; i < 10; i++) { cf[2 * i] = 0.f; cf[2 * i + 1] = 0.f; }
        }
    }
}

// 64 records of 80 bytes (one per lane, consecutive in memory from dst0) through the wave's LDS stage: a lane's 80 bytes as five
// 16-byte pieces would leave the wave as five stores of 64 scattered pieces each (every piece a partial line for the L2); staged,
// every store instruction writes 1024 consecutive bytes.  `nlive`: the live lanes are 0 .. nlive - 1.
__device__ __forceinline__ void store_records_80(int4 *__restrict__ dst0, int4 *__restrict__ stage /* 320 int4 of this wave */, const int4 (&rec)[5],
                                                 int lane, int nlive)
{
#pragma unroll
    for (int q = 0; q < 5; q++) stage[lane * 5 + q] = rec[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int i = q * 64 + lane;
        if (i < nlive * 5) dst0[i] = stage[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();             // (the stage is reused)
}

// Load descriptors for accumulate_grouped_kernel, 128 ints per record, laid out so that one coalesced load per wave
// brings them in lane-distributed: for component ig and node k
//   tab[4*ig + k]      = (row - row0)*pitch + kRowPad - first   (float index of trace sample 0 relative to the group base
//                                                        G + row0*pitch, row0 = first row of the cell's first node)
//   tab[64 + 4*ig + k] = (row - row0)*pitch             (clamp floor; ceiling = floor + pitch - 4)
//   tab[40 + ig]       = last stored sample of the blended trace (max over the nodes)
//   tab[50], tab[51]   = minimum of those over the horizontal (1-5, 9) / vertical (6-8, 10) components
//   tab[52], tab[53]   = minimum / maximum over the cell's rows of (kRowPad - first): where trace sample 0 sits inside its row
//                        (accumulate_cell_kernel: a tile that stays inside every row is loaded without clamps)
//   tab[54]            = 1 when the components of each node start at the same sample: the sample-0 positions of a node's rows are
//                        then `pitch` apart and the kernel takes four descriptors instead of forty
//   tab[64 + 40 + 2*i], [.. + 1] = wl, wr: per-component interpolation coefficients of THIS centroid
//       (sparse_trace.f90:643-647 with the factors of seismogram.f90:171-250): wl = (1 - w) * factor, wr = w * factor,
//       each rounded on its own, for the i-th component in application order 0 1 2 8 | 3 4 | 5 6 7 9 (ng = 8:
//       0 1 2 | 3 4 | 5 6 7).  Read by the grouped kernel with scalar loads, which takes them off the vector pipe.
// Every thread owns one 512-byte row.  All span look-ups come first, then each of the row's four 128-byte lines goes
// out as consecutive 16-byte stores, so that a line is complete in L2 before it leaves it (interleaving the stores
// with the look-ups left every line open for microseconds: partial-line write-backs, 0.32 ms per 1.3 M records).
template <int NG>
__device__ __forceinline__ bool write_tab(int *__restrict__ tb, float *__restrict__ cline, const GeoRec &g, const int2 *__restrict__ span, int pitch, float sd,
                                          bool full, const unsigned char *__restrict__ endz)
{
    bool all_endzero = true;          // returned: (full rows) every row of the cell ends in an exact zero -- no tail rule for this group
    // full: this centroid starts a group and the kernel reads the whole row; otherwise only its coefficients (they
    // share the row's last 128-byte line with the clamp floors of components 9 and 10, which are then not needed)
    const int nn = (g.flags & 1) ? 1 : 4;
    int bases[NG][4], floors[NG][4], jend[12];
    int jmin_h = 0x7fffffff, jmin_d = 0x7fffffff;
    int amin = 0x7fffffff, amax = -0x7fffffff;      // range of (trace sample 0 inside its row) over the cell's rows
    bool uni = true;                                // the components of every node start at the same sample (the usual database)
    if (full) {
    // Rows whose stored trace ends in an exact zero (the reference's trace_pack keeps one of the zeros that follow the last
    // non-zero sample, sparse_trace.f90:535,545, so this is the normal case for traces that die out inside the database's
    // time range): the repeated end value is 0 and the rule "factor * last after the span" (sparse_trace.f90:698-703) adds
    // the same signed zeros as the interpolation formula does -- the kernel may then skip its tail variant for this group.
    // endz[row]: the row's end value is zero (set by kiwi_hip_set_gfdb).
    bool endzero = true;
#pragma unroll
    for (int ig = 0; ig < NG; ig++) {
        int je = -0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = g.row[k < nn ? k : 0] + ig;
            const int2 sp = span[row];
            // offsets are relative to the first row of the cell's first node (g.row[0]): the kernels add them to a 64-bit
            // base, so a tensor of any size works as long as one cell spans less than 2^31 floats (checked by the host)
            bases[ig][k] = (row - g.row[0]) * pitch + kRowPad - sp.x;
            floors[ig][k] = (row - g.row[0]) * pitch;
            amin = min(amin, kRowPad - sp.x); amax = max(amax, kRowPad - sp.x);
            uni = uni && bases[ig][k] - floors[ig][k] == bases[0][k] - floors[0][k];
            if (k < nn) { je = max(je, sp.y); endzero = endzero && endz[row]; }
        }
        jend[ig] = je;
        const bool horiz = (ig <= 4) || (ig == 8);
        if (horiz) jmin_h = min(jmin_h, je); else jmin_d = min(jmin_d, je);
    }
#pragma unroll
    for (int ig = NG; ig < 10; ig++) jend[ig] = 0;
    jend[10] = endzero ? 0x7fffffff : jmin_h;
    jend[11] = endzero ? 0x7fffffff : jmin_d;
    all_endzero = endzero;
    }
    float cf[20];
    coef_line<NG>(g, sd, cf);
    int4 *t4 = reinterpret_cast<int4 *>(tb);
    if (full) {
#pragma unroll
        for (int ig = 0; ig < NG; ig++) t4[ig] = make_int4(bases[ig][0], bases[ig][1], bases[ig][2], bases[ig][3]);
#pragma unroll
        for (int q = 0; q < 3; q++) t4[10 + q] = make_int4(jend[4 * q], jend[4 * q + 1], jend[4 * q + 2], jend[4 * q + 3]);
        t4[13] = make_int4(amin, amax, uni ? 1 : 0, 0);
#pragma unroll
        for (int ig = 0; ig < NG; ig++) t4[16 + ig] = make_int4(floors[ig][0], floors[ig][1], floors[ig][2], floors[ig][3]);
    }
    float4 *f4 = reinterpret_cast<float4 *>(cline);           // (80-byte lines: 16-byte aligned)
#pragma unroll
    for (int q = 0; q < (NG == 10 ? 5 : 4); q++) f4[q] = make_float4(cf[4 * q], cf[4 * q + 1], cf[4 * q + 2], cf[4 * q + 3]);
    return all_endzero;
}

__global__ __launch_bounds__(256) void geometry_kernel(
    const float *__restrict__ cent, const int *__restrict__ cent_ofs, EvalParams ep, GfMeta gm,
    const int2 *__restrict__ span, const RecvDev *__restrict__ recv, GeoRec *__restrict__ out,
    int *__restrict__ tab, float *__restrict__ coefs /* [record][kCoefLine] interpolation coefficients, see coef_wl */, int *__restrict__ spanbuf, int *__restrict__ spansrc /* optional [source][receiver][kSpanInts] */,
    int *__restrict__ pairflag /* optional [source][receiver]: bit 0 some centroid of the pair is added in part (a trace is missing),
                                  bit 1 some centroid is left out (a trace missing or outside the database), bit 2 some group's rows do
                                  not all end in zero (the tail rule can apply); see cell_pair(), multi_taken() */,
    const unsigned char *__restrict__ endz /* per GF row: its end value is zero (write_tab) */,
    const int *__restrict__ synrow /* optional [source]: source whose synthetics this one shares; != own index: nothing to do */,
    int *__restrict__ lmax_out /* optional [source][receiver], preset to -1: index of the LAST centroid that reaches the rotated add of the
                                  rotating branch (see the span reduction below) */,
    const int *__restrict__ lmax_in /* optional: the same, from a pass in front of this one */,
    int4 *__restrict__ off4 /* optional [record]: compact load descriptors INSTEAD of the 512-byte rows of `tab` (see off4 below) */)
{
    const int s = blockIdx.y;
    if (synrow && synrow[s] != s) return;
    const int c0 = cent_ofs[ep.isrc0 + s], nc = cent_ofs[ep.isrc0 + s + 1] - c0;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nc * ep.nrec) return;                    // (whole workgroup)
    // lanes past the end of the source's records stay in the wave -- the span reduction below is a wave operation --
    // as copies of its last record that neither store nor count
    const bool live = idx < nc * ep.nrec;
    if (!live) { if (!spanbuf && !spansrc && !off4) return; idx = nc * ep.nrec - 1; }      // (off4: the staged stores below want whole waves)
    int rot_gap = -1, rot_rows[4] = { -1, -1, -1, -1 };   // rotating branch left at a missing horizontal trace: how many of its components went into the temporaries, and the rows
    const int r = idx / nc, c = idx - r * nc;
    const RecvDev &rv = recv[r];
    const float *ce = cent + (size_t)(c0 + c) * 10;
    GeoRec g;
    const float dnorth = ce[0], deast = ce[1], depth = ce[2], time = ce[3];
    const float pi_f = 3.14159265358979f;           // constants.f90:21
    const double pi_d = (double)pi_f;               // constants.f90:22
    const float earthradius = 6371.f * 1000.f;      // constants.f90:23

    // ---- approx_differential_azidist, orthodrome.f90:77-156 (exact branch; r == 0 -> const azimuth)
    double azi, bazi, dist;
    {
        const double azimuth = rv.azi0, backazimuth = rv.bazi0, d0 = rv.dist0;
        const float rr = sqrtf(dnorth * dnorth + deast * deast);
        const double rd = (double)rr;
        if (d0 / rd > 1.79769313486231570815e308) {
            azi = azimuth; bazi = backazimuth;
            dist = d0 - ((double)dnorth * cos(azimuth) + (double)deast * sin(azimuth));
        } else {
            const double a = rd / (double)earthradius;
            const double b = d0 / (double)earthradius;
            const double lambda = (double)libm32::atan2f_glibc(deast, dnorth);
            const double gamma = azimuth - lambda;
            const double ca = cos(a), cb = cos(b), sa = sin(a), sb = sin(b), sg = sin(gamma);
            const double cc = acos(clipd(ca * cb + sa * sb * cos(gamma), -1., 1.));
            const double sc = sin(cc), cosc = cos(cc);
            double alpha = asin(clipd(sa * sg / sc, -1., 1.));
            double beta = asin(clipd(sb * sg / sc, -1., 1.));
            if (ca - cb * cosc < 0) alpha = (alpha > 0) ? pi_d - alpha : -pi_d - alpha;
            if (cb - ca * cosc < 0) beta = (beta > 0) ? pi_d - beta : -pi_d - beta;
            dist = cc * (double)earthradius;
            bazi = wrapd(backazimuth + alpha, -pi_d, pi_d);
            azi = wrapd(lambda - pi_d - beta, -pi_d, pi_d);
        }
    }

    // ---- make_weights, seismogram.f90:316-336
    {
        const float azf = (float)azi;
        const float sa = sin32(azf), ca = cos32(azf), s2a = sin32(2.f * azf), c2a = cos32(2.f * azf);
        const float *m = ce + 4;
        g.f[0] = m[0] * (ca * ca) + m[1] * (sa * sa) + m[3] * s2a;
        g.f[1] = m[4] * ca + m[5] * sa;
        g.f[2] = m[2];
        g.f[3] = 0.5f * (m[1] - m[0]) * s2a + m[3] * c2a;
        g.f[4] = m[5] * ca - m[4] * sa;
        g.f[5] = m[0] * (sa * sa) + m[1] * (ca * ca) - m[3] * s2a;
    }

    // ---- time shift, seismogram.f90:139 + sparse_trace.f90:640-642
    {
        const float rshift = time / gm.dt;
        const float fl = floorf(rshift);
        g.ishift = (int)fl;
        g.wfrac = rshift - (float)g.ishift;
    }

    // ---- rotation of horizontals, seismogram.f90:159-165
    {
        const double lambda = bazi - rv.bazi0;
        g.flags = (lambda != 0.) ? 2 : 0;
        g.cl = (float)cos(lambda);
        g.sl = (float)sin(lambda);
    }

    // ---- GF indices, gfdb.f90:781-815
    const float x = (float)dist, z = depth - rv.depth;
    int ix0, iz0, ix1, iz1;
    float dix = 0.f, diz = 0.f;
    if (ep.bilinear) {
        ix0 = (int)floorf((x - gm.firstx) / (gm.dx * (float)ep.xus)) * ep.xus + 1;
        iz0 = (int)floorf((z - gm.firstz) / (gm.dz * (float)ep.zus)) * ep.zus + 1;
        ix1 = ix0 + ep.xus; iz1 = iz0 + ep.zus;
        dix = (x - gm.firstx - (float)(ix0 - 1) * gm.dx) / (gm.dx * (float)ep.xus);
        diz = (z - gm.firstz - (float)(iz0 - 1) * gm.dz) / (gm.dz * (float)ep.zus);
    } else {
        ix0 = (int)roundf((x - gm.firstx) / gm.dx) + 1;       // nint
        iz0 = (int)roundf((z - gm.firstz) / gm.dz) + 1;
        ix1 = ix0 + 1; iz1 = iz0 + 1;
    }
    const bool direct = (dix == 0.f && diz == 0.f);            // gfdb.f90:890
    if (direct) g.flags |= 1;
    // bit2: this centroid sits at exactly the same point as its predecessor in the table (the nt
    // time steps of one sub-fault, source_bilat.f90:443-457): same azimuth, distance, GF nodes and
    // blend weights, so the blended traces can be reused (accumulate_grouped_kernel)
    if (c > 0) {
        const float *pe = ce - 10;
        if (pe[0] == dnorth && pe[1] == deast && pe[2] == depth) g.flags |= 4;
    }
    g.w[0] = (1.f - dix) * (1.f - diz);
    g.w[1] = (1.f - dix) * diz;
    g.w[2] = dix * (1.f - diz);
    g.w[3] = dix * diz;
    auto inrange = [&](int ix, int iz) { return ix >= 1 && ix <= gm.nx && iz >= 1 && iz <= gm.nz; };
    auto rowof = [&](int ix, int iz) { return ((ix - 1) * gm.nz + (iz - 1)) * gm.ng; };
    bool ok = inrange(ix0, iz0);
    if (!direct) ok = ok && inrange(ix0, iz1) && inrange(ix1, iz0) && inrange(ix1, iz1);
    g.row[0] = g.row[1] = g.row[2] = g.row[3] = -1;
    int nlim_h = 0, nlim_d = 0;                    // components of the horizontal / vertical block that are added
    if (ok) {
        g.row[0] = rowof(ix0, iz0);
        if (!direct) { g.row[1] = rowof(ix0, iz1); g.row[2] = rowof(ix1, iz0); g.row[3] = rowof(ix1, iz1); }
        else { g.row[1] = g.row[2] = g.row[3] = g.row[0]; }
        // A trace that is not stored makes gfdb_get_trace[_bilin] return null (gfdb.f90:899-903,1003) and the reference
        // leaves the centroid AT THAT COMPONENT (`if (.not. associated(tracep)) cycle`, seismogram.f90:171-250): what was
        // added before stays, the rest -- including the vertical block when the gap is in the horizontal one -- is not
        // added.  In the rotate branch the horizontals are collected in temporaries that are only added after the last of
        // them (:196-203), so there a gap drops all of them.  Components in application order: 1 2 3 [9] 4 5 | 6 7 8 [10].
        const int nn = direct ? 1 : 4;
        const int nH = gm.ng == 10 ? 6 : 5, nD = gm.ng == 10 ? 4 : 3;
        auto stored = [&](int ig) {
            for (int k = 0; k < nn; k++) { const int2 sp = span[g.row[k] + ig]; if (sp.y < sp.x) return false; }
            return true;
        };
        bool hfull = true;
        if (ep.nogaps) {
            // every trace of the database is stored: all components of both blocks are added (what the loops below find after
            // forty span look-ups per record -- a third of this kernel's time at cfg3 once the descriptor rows were gone)
            nlim_h = rv.need_h ? nH : 0;
            nlim_d = rv.has_d ? nD : 0;
            if (rv.need_h && (g.flags & 2) && live && lmax_out) atomicMax(&lmax_out[(size_t)s * ep.nrec + r], c);
        } else {
        if (rv.need_h) {
            int k = 0;
            for (; k < nH; k++) if (!stored(gm.ng == 10 ? (k < 3 ? k : (k == 3 ? 8 : k - 1)) : k)) break;
            hfull = (k == nH);
            nlim_h = hfull ? nH : ((g.flags & 2) ? 0 : k);
            if (!hfull && (g.flags & 2)) { rot_gap = k; for (int q = 0; q < 4; q++) rot_rows[q] = g.row[q]; }
            if (hfull && (g.flags & 2) && live && lmax_out) atomicMax(&lmax_out[(size_t)s * ep.nrec + r], c);
        }
        if (rv.has_d && hfull) {
            int k = 0;
            for (; k < nD; k++) if (!stored(k < 3 ? 5 + k : 9)) break;
            nlim_d = k;
        }
        }
        const bool complete = (!rv.need_h || nlim_h == nH) && (!rv.has_d || nlim_d == nD);
        if (!complete) {
            if (nlim_h == 0 && nlim_d == 0) ok = false;                  // nothing of this centroid is added
            else { g.flags |= 8 | (nlim_h << 8) | (nlim_d << 12); if (pairflag) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 1); }
        }
        if (!ok) g.row[0] = -1;
    }
    if (pairflag && live && g.row[0] < 0) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 2);
    // group hint (used by accumulate_grouped_kernel when a group STARTS at this centroid): how many
    // following centroids sit at this same point with their integer shifts within the LDS halo, and
    // the spread of those shifts.  pad = len | (smax - ishift) << 8 | (ishift - smin) << 16
    {
        int len = 1, smin = g.ishift, smax = g.ishift;
        if (g.row[0] >= 0 && !ep.cellmode) len = group_len(cent, c0, nc, c, gm.dt, smin, smax);
        g.pad = len | ((smax - g.ishift) << 8) | ((g.ishift - smin) << 16);
    }
    // natural span of the synthetic strips (seismogram.f90:102-130 + sparse_trace.f90:648-668): union over
    // centroids of [first + shift, last + shift + 1], horizontals (all share one span, :196-197) and vertical
    // separately; only needed to size the comparator's FFT (comparator.f90:464-486)
    if (spanbuf || spansrc) {
        int lo_1 = 0x7fffffff, hi_1 = -0x7fffffff, lo_2 = 0x7fffffff, hi_2 = -0x7fffffff, lo_d = 0x7fffffff, hi_d = -0x7fffffff;
        if (live && g.row[0] >= 0) {
            const int nn = (g.flags & 1) ? 1 : 4;
            const int nH1 = gm.ng == 10 ? 4 : 3;          // components of the radial sum
            for (int i = 0; i < nlim_h + nlim_d; i++) {   // the components that are added (all of the needed ones, normally)
                const bool horiz = i < nlim_h;
                const int q = horiz ? i : i - nlim_h;
                const int ig = horiz ? (gm.ng == 10 ? (q < 3 ? q : (q == 3 ? 8 : q - 1)) : q) : (q < 3 ? 5 + q : 9);
                int lo = 0x7fffffff, hi = -0x7fffffff;
                for (int k = 0; k < nn; k++) { const int2 sp = span[g.row[k] + ig]; lo = min(lo, sp.x); hi = max(hi, sp.y); }
                if (!horiz) { lo_d = min(lo_d, lo); hi_d = max(hi_d, hi); }
                else if (q < nH1) { lo_1 = min(lo_1, lo); hi_1 = max(hi_1, hi); }
                else { lo_2 = min(lo_2, lo); hi_2 = max(hi_2, hi); }
            }
            if ((g.flags & 2) && nlim_h) {                // rotating branch: both sums get the union before the rotated add
                lo_1 = lo_2 = min(lo_1, lo_2); hi_1 = hi_2 = max(hi_1, hi_2);
            }
            // strip spans of this centroid's contribution: [first + shift, last + shift + 1]; empty stays (+inf, -inf)
            if (hi_1 >= lo_1) { lo_1 += g.ishift; hi_1 += g.ishift + 1; }
            if (hi_2 >= lo_2) { lo_2 += g.ishift; hi_2 += g.ishift + 1; }
            if (nlim_d && hi_d >= lo_d) { lo_d += g.ishift; hi_d += g.ishift + 1; } else { lo_d = 0x7fffffff; hi_d = -0x7fffffff; }
        } else if (live && rot_gap > 0 && lmax_in && c < lmax_in[(size_t)s * ep.nrec + r]) {
            // The rotating branch collects a centroid's horizontals in temporaries that KEEP their extent from centroid to centroid
            // (seismogram.f90:160-190: `displacement_temp(:)%data(:) = 0.`); a centroid that leaves at a missing trace has extended
            // them by what it had added so far, and the next centroid that reaches the rotated add passes that extent on to the
            // strips (strip_extend_to_same_span_4, :193).  So the components in front of the gap count for the strips' spans -- if a
            // later centroid of this (source, receiver) completes its horizontals.
            const int nn = (g.flags & 1) ? 1 : 4;
            for (int i = 0; i < rot_gap; i++) {
                const int ig = gm.ng == 10 ? (i < 3 ? i : (i == 3 ? 8 : i - 1)) : i;
                for (int k = 0; k < nn; k++) { const int2 sp = span[rot_rows[k] + ig]; lo_1 = min(lo_1, sp.x); hi_1 = max(hi_1, sp.y); }
            }
            lo_1 += g.ishift; hi_1 += g.ishift + 1;
            lo_2 = lo_1; hi_2 = hi_1;
        }
        // Union over the centroids of one (source, receiver): the lanes of a wave that belong to the same receiver are
        // consecutive (idx = r * nc + c), so a segmented suffix reduction leaves the union of each run in its first lane
        // and only that lane goes to memory -- a 64th of the atomics (135 centroids updating the same six words made this
        // kernel 4 x slower per record at cfg5 than at cfg3).
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int kr = __shfl_down(r, off, 64);
            const int a1 = __shfl_down(lo_1, off, 64), b1 = __shfl_down(hi_1, off, 64);
            const int a2 = __shfl_down(lo_2, off, 64), b2 = __shfl_down(hi_2, off, 64);
            const int ad = __shfl_down(lo_d, off, 64), bd = __shfl_down(hi_d, off, 64);
            if (lane + off < 64 && kr == r) {
                lo_1 = min(lo_1, a1); hi_1 = max(hi_1, b1);
                lo_2 = min(lo_2, a2); hi_2 = max(hi_2, b2);
                lo_d = min(lo_d, ad); hi_d = max(hi_d, bd);
            }
        }
        const int rprev = __shfl_up(r, 1, 64);
        if (lane == 0 || rprev != r) {
            if (spanbuf) {                                // per receiver over all sources: [horizontal lo, hi, vertical lo, hi]
                if (max(hi_1, hi_2) >= min(lo_1, lo_2)) { atomicMin(&spanbuf[4 * r + 0], min(lo_1, lo_2)); atomicMax(&spanbuf[4 * r + 1], max(hi_1, hi_2)); }
                if (hi_d >= lo_d) { atomicMin(&spanbuf[4 * r + 2], lo_d); atomicMax(&spanbuf[4 * r + 3], hi_d); }
            }
            if (spansrc) {                               // the same per trial source and strip: data spans of ITS synthetic strips
                int *sp = spansrc + ((size_t)s * ep.nrec + r) * kSpanInts;
                if (hi_1 >= lo_1) { atomicMin(&sp[0], lo_1); atomicMax(&sp[1], hi_1); }
                if (hi_2 >= lo_2) { atomicMin(&sp[2], lo_2); atomicMax(&sp[3], hi_2); }
                if (hi_d >= lo_d) { atomicMin(&sp[4], lo_d); atomicMax(&sp[5], hi_d); }
            }
        }
    }
    if (!out) return;
    if (!live && !off4) return;
    const size_t base = (size_t)(c0 - cent_ofs[ep.isrc0]) * ep.nrec + (size_t)r * nc + c;
    if (off4) {
        // Compact path: records and coefficient lines leave the wave as whole kilobytes (store_records_80).  The lanes still here
        // are the live ones, a prefix of the wave, and their records are consecutive in memory (idx = r nc + c).
        __shared__ int4 stage_all[4 * 320];
        const int lane = threadIdx.x & 63;
        int4 *stage = stage_all + (threadIdx.x >> 6) * 320;
        const int nlive = __popcll(__builtin_amdgcn_ballot_w64(live));     // (the other lanes hold copies of the last record: staged, not stored)
        const unsigned b_lo = __builtin_amdgcn_readfirstlane((unsigned)base), b_hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const size_t base0 = ((size_t)b_hi << 32) | b_lo;
        int4 rec[5];
        __builtin_memcpy(rec, &g, 80);
        store_records_80(reinterpret_cast<int4 *>(out + base0), stage, rec, lane, nlive);
        if (tab) {
            float cf[20];
#pragma unroll
            for (int i = 0; i < 20; i++) cf[i] = 0.f;
            int4 o = make_int4(0, 0, 0, 0);
            if (g.row[0] >= 0) {
                // Compact descriptors (databases whose components of a node all start at the same sample and whose rows all end in an
                // exact zero -- what a database reader delivers for traces that die out inside the time range; the host checks it once,
                // kiwi_hip_set_gfdb): everything a kernel takes from a 512-byte descriptor row then follows from the record's four node
                // rows and FOUR numbers, the position of trace sample 0 inside the rows of each node -- 16 bytes per record, written by
                // every thread (coalesced) instead of 432 bytes by the fifth of the lanes that sit at a group start (those rows were
                // 2.1 GB per 4096 cfg3 sources and 1.8 of this kernel's 4.1 ms).  The kernels rebuild the lane-distributed row from
                // them (desc_expand, kiwi_accum.inc).
                const int nn = (g.flags & 1) ? 1 : 4;
                int ok[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ok[k] = kRowPad - span[g.row[k < nn ? k : 0]].x;
                o = make_int4(ok[0], ok[1], ok[2], ok[3]);
                if (gm.ng == 10) coef_line<10>(g, rv.sd, cf); else coef_line<8>(g, rv.sd, cf);
            }
            if (live) off4[base] = o;
            __builtin_memcpy(rec, cf, 80);
            store_records_80(reinterpret_cast<int4 *>(coefs + base0 * kCoefLine), stage, rec, lane, nlive);
        }
        return;
    }
    out[base] = g;
    if (tab && g.row[0] >= 0) {
        // cell mode: only the coefficient line here, cellgroup_kernel completes the rows of the group starts it finds
        const bool full = !ep.cellmode && (!(g.flags & 4) || starts_group(cent, c0, nc, c, gm.dt));
        bool ez;
        if (gm.ng == 10) ez = write_tab<10>(tab + base * 128, coefs + base * kCoefLine, g, span, gm.pitch, rv.sd, full, endz);
        else ez = write_tab<8>(tab + base * 128, coefs + base * kCoefLine, g, span, gm.pitch, rv.sd, full, endz);
        if (pairflag && !ez) atomicOr(&pairflag[(size_t)s * ep.nrec + r], 4);
    }
}

// Second geometry pass of the cell mode: consecutive centroids (table order) of one (source, receiver) whose four GF nodes
// are the same -- neighbouring sub-faults of a rupture are hundreds of metres apart, the nodes kilometres -- form a
// group: accumulate_cell_kernel fetches the raw node traces ONCE per group and blends them per centroid with that
// centroid's weights.  Groups are cut greedily from the start of a same-cell run (length <= kMaxGroup, integer shifts
// within the LDS halo).  Pairs accumulate_cell_kernel does not take (see cell_pair) get same-point groups.  Thread per record: a thread
// that finds itself at a group start leaves the hint in its record and completes its descriptor row.
__device__ __forceinline__ bool same_cell(const GeoRec *__restrict__ a, const int (&row)[4])
{
    const int4 r = *reinterpret_cast<const int4 *>(a->row);
    return r.x == row[0] && r.y == row[1] && r.z == row[2] && r.w == row[3];
}


__global__ __launch_bounds__(256) void cellgroup_kernel(const int *__restrict__ cent_ofs, EvalParams ep, GfMeta gm,
                                                        const int2 *__restrict__ span, const RecvDev *__restrict__ recv,
                                                        GeoRec *__restrict__ recs, int *__restrict__ tab, float *__restrict__ coefs,
                                                        const int *__restrict__ pairflag, const unsigned char *__restrict__ endz,
                                                        const int *__restrict__ synrow, int cell_range /* largest shift range of a cell group */,
                                                        int compact /* descriptors come from off4 (geometry_kernel): no rows to complete */)
{
    const int s = blockIdx.y;
    if (synrow && synrow[s] != s) return;
    const int cb = cent_ofs[ep.isrc0], c0 = cent_ofs[ep.isrc0 + s], nc = cent_ofs[ep.isrc0 + s + 1] - c0;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * ep.nrec) return;
    const int r = idx / nc, c = idx - r * nc;
    const size_t base0 = (size_t)(c0 - cb) * ep.nrec + (size_t)r * nc;
    GeoRec *__restrict__ rr = recs + base0;
    const GeoRec me = rr[c];
    if (me.row[0] < 0) return;
    int len = 1, smin = me.ishift, smax = me.ishift;
    {
        // cell pairs: a group runs on while the four nodes stay the same; other pairs: while the POINT stays the same
        // (flags bit 2 of the follower), which is what accumulate_grouped_kernel's shared blended tile needs
        const bool cellp = cell_pair(recv[r], pairflag, s, ep.nrec, r);
        int r0 = c;
        while (r0 > 0 && same_cell(rr + r0 - 1, me.row) && (cellp || (rr[r0].flags & 4))) r0--;
        int pos = r0;
        for (;;) {
            len = 1;
            smin = smax = rr[pos].ishift;
            for (int k = pos + 1; k < nc && len < kMaxGroup; k++) {
                if (!same_cell(rr + k, me.row) || !(cellp || (rr[k].flags & 4))) break;
                const int sh = rr[k].ishift;
                const int nmin = min(smin, sh), nmax = max(smax, sh);
                if (nmax - nmin > (cellp ? cell_range : kHalo - 10)) break;
                smin = nmin; smax = nmax; len++;
            }
            if (pos == c) break;                  // this centroid starts a group
            if (pos + len > c) return;            // inside a group: its coefficient line is all the kernel reads of its row
            pos += len;
        }
    }
    rr[c].pad = len | ((smax - me.ishift) << 8) | ((me.ishift - smin) << 16);
    if (compact) return;
    if (gm.ng == 10) write_tab<10>(tab + (base0 + c) * 128, coefs + (base0 + c) * kCoefLine, me, span, gm.pitch, recv[r].sd, true, endz);
    else write_tab<8>(tab + (base0 + c) * 128, coefs + (base0 + c) * kCoefLine, me, span, gm.pitch, recv[r].sd, true, endz);
}

} // namespace kiwi
