/* kiwi_gfdb_hdf5.c -- reader (and, for tests and conversions, writer) of Kiwi's on-disk Green's function database
 * through the HDF5 C library.  Layout restated from the reference's Fortran I/O layer (gfdb_io_hdf.f90):
 *
 *   <base>.index            scalar datasets dt, dx, dz, firstx, firstz (real), nchunks, nx, nxc, nz, ng (integer)
 *                           (:118-178,181-236; firstx/firstz may be absent in old files -> 0)
 *   <base>.<ichunk>.chunk   ichunk = 1..nchunks, each holding nxc receiver distances (the last one the rest,
 *                           gfdb.f90:250-257):
 *       /index              H5T_STD_REF_OBJ, Fortran dims (ng, nz, nxc): object reference of every stored trace
 *                           (:262-289,396-414); a zero reference = trace not stored
 *       /gf/<ixc>/<iz>/<ig> 1-D real dataset: the trace's strips packed back to back, with integer attributes
 *                           pofs(nstrips) = 1-based start of every strip in the packed array, ofs(nstrips) = sample
 *                           index of every strip's first sample (:341-392; sparse_trace.f90:795-877)
 *
 * A trace unpacks to: zeros in the gaps between strips, span = [ofs(1), ofs(n) + len(n) - 1].  The database is
 * turned into the dense arrays kiwi_hip_set_gfdb takes.  Kept OUT of libkiwi_hip.so so that the product library does
 * not depend on HDF5. */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float dt, dx, dz, firstx, firstz;
    int nchunks, nx, nxc, nz, ng;
} kiwi_gfdb_index;

static int fail(char *err, int errlen, const char *fmt, const char *arg)
{
    if (err && errlen > 0) snprintf(err, (size_t)errlen, fmt, arg);
    return 1;
}

static int read_scalar(hid_t file, const char *name, hid_t type, void *out)
{
    if (H5Lexists(file, name, H5P_DEFAULT) <= 0) return 1;
    hid_t d = H5Dopen2(file, name, H5P_DEFAULT);
    if (d < 0) return 1;
    const herr_t e = H5Dread(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, out);
    H5Dclose(d);
    return e < 0;
}

static int write_scalar(hid_t file, const char *name, hid_t type, const void *v)
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t d = H5Dcreate2(file, name, type, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    herr_t e = d < 0 ? -1 : H5Dwrite(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
    if (d >= 0) H5Dclose(d);
    H5Sclose(sp);
    return e < 0;
}

/* gfdb_io_read_index, gfdb_io_hdf.f90:118-178 */
int kiwi_gfdb_read_index(const char *base, kiwi_gfdb_index *ix, char *err, int errlen)
{
    char fn[4096];
    snprintf(fn, sizeof(fn), "%s.index", base);
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(fn, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return fail(err, errlen, "gfdb: failed to open file: %s", fn);
    memset(ix, 0, sizeof(*ix));
    int bad = read_scalar(f, "dt", H5T_NATIVE_FLOAT, &ix->dt) | read_scalar(f, "dx", H5T_NATIVE_FLOAT, &ix->dx) |
              read_scalar(f, "dz", H5T_NATIVE_FLOAT, &ix->dz);
    read_scalar(f, "firstx", H5T_NATIVE_FLOAT, &ix->firstx);          /* optional, :151-158 */
    read_scalar(f, "firstz", H5T_NATIVE_FLOAT, &ix->firstz);
    bad |= read_scalar(f, "nchunks", H5T_NATIVE_INT, &ix->nchunks) | read_scalar(f, "nx", H5T_NATIVE_INT, &ix->nx) |
           read_scalar(f, "nxc", H5T_NATIVE_INT, &ix->nxc) | read_scalar(f, "nz", H5T_NATIVE_INT, &ix->nz) |
           read_scalar(f, "ng", H5T_NATIVE_INT, &ix->ng);
    H5Fclose(f);
    if (bad) return fail(err, errlen, "gfdb: failed to read dataset from file: %s", fn);
    if (ix->nchunks < 1 || ix->nx < 1 || ix->nxc < 1 || ix->nz < 1 || ix->ng < 1)
        return fail(err, errlen, "gfdb: inconsistent index in file: %s", fn);
    return 0;
}

static int read_int_attr(hid_t d, const char *name, int **out, int *n)
{
    hid_t a = H5Aopen(d, name, H5P_DEFAULT);
    if (a < 0) return 1;
    hid_t sp = H5Aget_space(a);
    const hssize_t np = H5Sget_simple_extent_npoints(sp);
    H5Sclose(sp);
    *out = (int *)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
    *n = (int)np;
    const herr_t e = H5Aread(a, H5T_NATIVE_INT, *out);
    H5Aclose(a);
    return e < 0;
}

/* Walks every stored trace.  pass 0 (G == NULL): fills first/nsamp and *lmax.  pass 1: also the dense samples,
 * G[((ix*nz + iz)*ng + ig)*L + k], zero padded.  Traces are found through the /index references as the reference
 * does (chunk_open / gfdb_io_get_trace, gfdb_io_hdf.f90:417-520). */
int kiwi_gfdb_read_dense(const char *base, int L, float *G, int *first, int *nsamp, int *lmax, char *err, int errlen)
{
    kiwi_gfdb_index ix;
    if (kiwi_gfdb_read_index(base, &ix, err, errlen)) return 1;
    const size_t ntr = (size_t)ix.nx * ix.nz * ix.ng;
    for (size_t i = 0; i < ntr; i++) { first[i] = 0; nsamp[i] = 0; }
    if (G) memset(G, 0, ntr * (size_t)L * sizeof(float));
    int longest = 0;
    for (int ichunk = 1; ichunk <= ix.nchunks; ichunk++) {
        int nxcthis = ix.nxc;
        if (ichunk == ix.nchunks) nxcthis = ix.nx - (ichunk - 1) * ix.nxc;          /* gfdb.f90:252-253 */
        char fn[4096];
        snprintf(fn, sizeof(fn), "%s.%d.chunk", base, ichunk);
        hid_t f = H5Fopen(fn, H5F_ACC_RDONLY, H5P_DEFAULT);
        if (f < 0) return fail(err, errlen, "gfdb: failed to open file: %s", fn);
        hid_t di = H5Dopen2(f, "index", H5P_DEFAULT);
        if (di < 0) { H5Fclose(f); return fail(err, errlen, "gfdb: no index dataset in file: %s", fn); }
        const size_t nref = (size_t)nxcthis * ix.nz * ix.ng;
        hobj_ref_t *refs = (hobj_ref_t *)calloc(nref, sizeof(hobj_ref_t));
        hid_t sp = H5Dget_space(di);
        const hssize_t np = H5Sget_simple_extent_npoints(sp);
        H5Sclose(sp);
        if ((size_t)np != nref || H5Dread(di, H5T_STD_REF_OBJ, H5S_ALL, H5S_ALL, H5P_DEFAULT, refs) < 0) {
            free(refs); H5Dclose(di); H5Fclose(f);
            return fail(err, errlen, "gfdb: failed to read index dataset of file: %s", fn);
        }
        /* Fortran dims (ng, nz, nxc) = C order [ixc][iz][ig] */
        for (int ixc = 0; ixc < nxcthis; ixc++)
            for (int iz = 0; iz < ix.nz; iz++)
                for (int ig = 0; ig < ix.ng; ig++) {
                    const hobj_ref_t r = refs[((size_t)ixc * ix.nz + iz) * ix.ng + ig];
                    if (r == 0) continue;
                    hid_t d = H5Rdereference2(di, H5P_DEFAULT, H5R_OBJECT, &r);
                    if (d < 0) continue;
                    hid_t dsp = H5Dget_space(d);
                    const int npacked = (int)H5Sget_simple_extent_npoints(dsp);
                    H5Sclose(dsp);
                    int *pofs = NULL, *ofs = NULL, n1 = 0, n2 = 0;
                    float *packed = (float *)malloc(sizeof(float) * (size_t)(npacked > 0 ? npacked : 1));
                    int bad = read_int_attr(d, "pofs", &pofs, &n1) | read_int_attr(d, "ofs", &ofs, &n2);
                    if (!bad && npacked > 0) bad = H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, packed) < 0;
                    H5Dclose(d);
                    if (bad || n1 != n2 || n1 < 1 || npacked < 1) {
                        free(packed); free(pofs); free(ofs); free(refs); H5Dclose(di); H5Fclose(f);
                        return fail(err, errlen, "gfdb: failed to read a trace from file: %s", fn);
                    }
                    /* the strip table comes from the file: a truncated or corrupt chunk must end in the error above, not in
                     * a copy outside the buffers -- 1 <= pofs <= npacked, pofs and ofs increasing, strips not overlapping */
                    int okstrips = 1;
                    for (int s = 0; s < n1 && okstrips; s++) {
                        if (pofs[s] < 1 || pofs[s] > npacked) okstrips = 0;
                        if (s > 0 && (pofs[s] <= pofs[s - 1] || ofs[s] < ofs[s - 1] + (pofs[s] - pofs[s - 1]))) okstrips = 0;
                    }
                    if (!okstrips) {
                        free(packed); free(pofs); free(ofs); free(refs); H5Dclose(di); H5Fclose(f);
                        return fail(err, errlen, "gfdb: failed to read a trace from file: %s", fn);
                    }
                    /* trace_from_storable, sparse_trace.f90:849-877 */
                    const int gx = (ichunk - 1) * ix.nxc + ixc;
                    const size_t t = ((size_t)gx * ix.nz + iz) * ix.ng + ig;
                    const int lastlen = npacked - pofs[n1 - 1] + 1;
                    const int span0 = ofs[0], span1 = ofs[n1 - 1] + lastlen - 1;
                    first[t] = span0;
                    nsamp[t] = span1 - span0 + 1;
                    if (nsamp[t] > longest) longest = nsamp[t];
                    if (G) {
                        if (nsamp[t] > L) {
                            free(packed); free(pofs); free(ofs); free(refs); H5Dclose(di); H5Fclose(f);
                            return fail(err, errlen, "gfdb: trace longer than the dense row length in file: %s", fn);
                        }
                        for (int s = 0; s < n1; s++) {
                            const int n = (s + 1 < n1) ? pofs[s + 1] - pofs[s] : lastlen;
                            memcpy(G + t * (size_t)L + (size_t)(ofs[s] - span0), packed + (pofs[s] - 1), sizeof(float) * (size_t)n);
                        }
                    }
                    free(packed); free(pofs); free(ofs);
                }
        free(refs);
        H5Dclose(di);
        H5Fclose(f);
    }
    if (lmax) *lmax = longest;
    return 0;
}

/* ---------------------------------------------------------------- writer (tests, conversions)
 * gfdb_io_create_index / gfdb_io_create_chunk / gfdb_io_save_trace, gfdb_io_hdf.f90:181-414.  Strips are given
 * packed: for trace t (same linear order as above) packed samples pk[pk_ofs[t] .. pk_ofs[t+1]) and strips
 * st_ofs[t] .. st_ofs[t+1] with pofs (1-based into the trace's packed array) and ofs (sample index). */
int kiwi_gfdb_write(const char *base, const kiwi_gfdb_index *ix, const float *pk, const long long *pk_ofs,
                    const int *pofs, const int *ofs, const long long *st_ofs, char *err, int errlen)
{
    char fn[4096];
    snprintf(fn, sizeof(fn), "%s.index", base);
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return fail(err, errlen, "gfdb: failed to create file: %s", fn);
    int bad = write_scalar(f, "dt", H5T_NATIVE_FLOAT, &ix->dt) | write_scalar(f, "dx", H5T_NATIVE_FLOAT, &ix->dx) |
              write_scalar(f, "dz", H5T_NATIVE_FLOAT, &ix->dz) | write_scalar(f, "firstx", H5T_NATIVE_FLOAT, &ix->firstx) |
              write_scalar(f, "firstz", H5T_NATIVE_FLOAT, &ix->firstz) | write_scalar(f, "nchunks", H5T_NATIVE_INT, &ix->nchunks) |
              write_scalar(f, "nx", H5T_NATIVE_INT, &ix->nx) | write_scalar(f, "nxc", H5T_NATIVE_INT, &ix->nxc) |
              write_scalar(f, "nz", H5T_NATIVE_INT, &ix->nz) | write_scalar(f, "ng", H5T_NATIVE_INT, &ix->ng);
    H5Fclose(f);
    if (bad) return fail(err, errlen, "gfdb: failed to write dataset to file: %s", fn);
    for (int ichunk = 1; ichunk <= ix->nchunks; ichunk++) {
        int nxcthis = ix->nxc;
        if (ichunk == ix->nchunks) nxcthis = ix->nx - (ichunk - 1) * ix->nxc;
        snprintf(fn, sizeof(fn), "%s.%d.chunk", base, ichunk);
        f = H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        if (f < 0) return fail(err, errlen, "gfdb: failed to create file: %s", fn);
        hsize_t dims[3] = { (hsize_t)nxcthis, (hsize_t)ix->nz, (hsize_t)ix->ng };    /* Fortran (ng, nz, nxc) */
        hid_t sp = H5Screate_simple(3, dims, NULL);
        hid_t di = H5Dcreate2(f, "index", H5T_STD_REF_OBJ, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        const size_t nref = (size_t)nxcthis * ix->nz * ix->ng;
        hobj_ref_t *refs = (hobj_ref_t *)calloc(nref, sizeof(hobj_ref_t));
        hid_t ggf = H5Gcreate2(f, "gf", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        for (int ixc = 0; ixc < nxcthis && !bad; ixc++) {
            char name[64];
            snprintf(name, sizeof(name), "%d", ixc + 1);
            hid_t gx = -1;
            for (int iz = 0; iz < ix->nz && !bad; iz++) {
                hid_t gz = -1;
                for (int ig = 0; ig < ix->ng && !bad; ig++) {
                    const int gxi = (ichunk - 1) * ix->nxc + ixc;
                    const size_t t = ((size_t)gxi * ix->nz + iz) * ix->ng + ig;
                    const int ns = (int)(st_ofs[t + 1] - st_ofs[t]);
                    if (ns == 0) continue;
                    if (gx < 0) gx = H5Gcreate2(ggf, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
                    if (gz < 0) { char zn[64]; snprintf(zn, sizeof(zn), "%d", iz + 1); gz = H5Gcreate2(gx, zn, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); }
                    char gn[64];
                    snprintf(gn, sizeof(gn), "%d", ig + 1);
                    hsize_t np = (hsize_t)(pk_ofs[t + 1] - pk_ofs[t]), nst = (hsize_t)ns;
                    hid_t dsp = H5Screate_simple(1, &np, NULL), asp = H5Screate_simple(1, &nst, NULL);
                    hid_t d = H5Dcreate2(gz, gn, H5T_NATIVE_FLOAT, dsp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
                    hid_t a1 = H5Acreate2(d, "pofs", H5T_NATIVE_INT, asp, H5P_DEFAULT, H5P_DEFAULT);
                    hid_t a2 = H5Acreate2(d, "ofs", H5T_NATIVE_INT, asp, H5P_DEFAULT, H5P_DEFAULT);
                    bad |= d < 0 || a1 < 0 || a2 < 0;
                    if (!bad) {
                        bad |= H5Awrite(a1, H5T_NATIVE_INT, pofs + st_ofs[t]) < 0;
                        bad |= H5Awrite(a2, H5T_NATIVE_INT, ofs + st_ofs[t]) < 0;
                        bad |= H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, pk + pk_ofs[t]) < 0;
                        bad |= H5Rcreate(&refs[((size_t)ixc * ix->nz + iz) * ix->ng + ig], gz, gn, H5R_OBJECT, -1) < 0;
                    }
                    if (a1 >= 0) H5Aclose(a1);
                    if (a2 >= 0) H5Aclose(a2);
                    if (d >= 0) H5Dclose(d);
                    H5Sclose(dsp); H5Sclose(asp);
                }
                if (gz >= 0) H5Gclose(gz);
            }
            if (gx >= 0) H5Gclose(gx);
        }
        if (!bad) bad |= H5Dwrite(di, H5T_STD_REF_OBJ, H5S_ALL, H5S_ALL, H5P_DEFAULT, refs) < 0;
        free(refs);
        H5Gclose(ggf); H5Dclose(di); H5Sclose(sp); H5Fclose(f);
        if (bad) return fail(err, errlen, "gfdb: failed to write file: %s", fn);
    }
    return 0;
}
