/*
 * ko_eikonal.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates the variable-rupture-speed sources:
 *   source_eikonal.f90 / source_mt_eikonal.f90 (identical up to parameter positions and where
 *   the moment tensor comes from), eikonal.f90 (fast marching), heap.f90 (index heap with back
 *   pointers), geometry.f90 (circle -> polygon, half-space clipping), and the two crust2x2.f90
 *   functions that act on an already looked-up 1-D profile.
 * The CRUST2.0 table lookup itself (crust2x2.f90:90-105, data files under aux/) is I/O and is NOT
 * restated: the caller supplies the profiles (tests fetch them from the reference build).
 * All arithmetic is default real, in the reference's order.
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float PI_F = 3.14159265358979f;

/* dot_product / matmul on 3-vectors, summed 1,2,3 */
static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void matvec(float m[3][3], const float v[3], float o[3])
{
    for (int i = 0; i < 3; i++) o[i] = (m[i][0] * v[0] + m[i][1] * v[1]) + m[i][2] * v[2];
}
static void matTvec(float m[3][3], const float v[3], float o[3])
{
    for (int i = 0; i < 3; i++) o[i] = (m[0][i] * v[0] + m[1][i] * v[1]) + m[2][i] * v[2];
}

/* crust2x2.f90:146-168 */
void ko_crust_profile_averages(const ko_crust_profile *p, float *vvp, float *vvs, float *vrho, float *vthi)
{
    float thi = 0.f, vp = 0.f, vs = 0.f, rho = 0.f;
    for (int i = 1; i < 7; i++) {            /* layers 2..7 (ice .. lower crust) */
        thi = thi + p->thickness[i];
        vp = vp + p->thickness[i] / p->vp[i];
        vs = vs + p->thickness[i] / p->vs[i];
        rho = rho + p->thickness[i] * p->rho[i];
    }
    *vvp = thi / vp; *vvs = thi / vs; *vrho = rho / thi; *vthi = thi;
}

/* crust2x2.f90:170-195 */
static void crust_at_depth(const ko_crust_profile *p, float depth, float *vp, float *vs, float *rho)
{
    float d = 0.f;
    for (int i = 2; i < 7; i++) {            /* layers 3..7 */
        d = d + p->thickness[i];
        if (d >= depth) { *vp = p->vp[i]; *vs = p->vs[i]; *rho = p->rho[i]; return; }
    }
    *vp = p->vp[7]; *vs = p->vs[7]; *rho = p->rho[7];
}

/* ---------------------------------------------------------------- geometry.f90 */
typedef struct { float point[3], normal[3]; } halfspace;
typedef struct { int n; float (*p)[3]; } polygon;

static int point_in_halfspace(const float pt[3], const halfspace *h)       /* :55-64 */
{
    float d[3] = { h->point[0] - pt[0], h->point[1] - pt[1], h->point[2] - pt[2] };
    return dot3(h->normal, d) >= 0.f;
}

/* geometry.f90:66-118 */
static void get_piercingpoint(const float a[3], const float b[3], const halfspace *h, float pp[3],
                              int *between_ab, int *parallel, int *a_inside_, int *b_inside_)
{
    float ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    float da[3] = { h->point[0] - a[0], h->point[1] - a[1], h->point[2] - a[2] };
    float db[3] = { h->point[0] - b[0], h->point[1] - b[1], h->point[2] - b[2] };
    float lambda_a = dot3(h->normal, da), lambda_b = dot3(h->normal, db), lambda_ab = dot3(h->normal, ab);
    int a_inside = lambda_a >= 0.f, b_inside = lambda_b >= 0.f;
    if (a_inside_) *a_inside_ = a_inside;
    if (b_inside_) *b_inside_ = b_inside;
    *between_ab = (a_inside && !b_inside) || (b_inside && !a_inside);
    *parallel = (lambda_ab * lambda_ab < dot3(ab, ab) / 16777216.f);       /* 2**digits(real) */
    if (*parallel && *between_ab) {
        const float *src = (fabsf(lambda_a) <= fabsf(lambda_b)) ? a : b;
        pp[0] = src[0]; pp[1] = src[1]; pp[2] = src[2];
        return;
    }
    if (*parallel && !*between_ab) { pp[0] = pp[1] = pp[2] = 0.f; return; }
    for (int i = 0; i < 3; i++) pp[i] = a[i] + ab[i] * lambda_a / lambda_ab;
}

/* geometry.f90:173-190 */
static void circle_to_polygon(const float center[3], float transform[3][3], int npoints, polygon *poly)
{
    poly->n = npoints;
    poly->p = (float(*)[3])malloc(sizeof(float[3]) * (size_t)(npoints > 0 ? npoints : 1));
    for (int i = 1; i <= npoints; i++) {
        float v[3] = { cosf((float)i * 2.f * PI_F / (float)npoints), sinf((float)i * 2.f * PI_F / (float)npoints), 0.f };
        float o[3];
        matvec(transform, v, o);
        for (int k = 0; k < 3; k++) poly->p[i - 1][k] = o[k] + center[k];
    }
}

/* geometry.f90:192-237 */
static void trim_polygon_one(const polygon *in, const halfspace *h, polygon *out)
{
    int n = in->n;
    float (*pp)[3] = (float(*)[3])malloc(sizeof(float[3]) * (size_t)(n > 0 ? n : 1));
    int *pierce = (int *)malloc(sizeof(int) * (size_t)(2 * n + 2)), *inside = pierce + n;
    int nt = 0;
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n, parallel, b_in;
        get_piercingpoint(in->p[i], in->p[j], h, pp[i], &pierce[i], &parallel, &inside[i], &b_in);
        if (inside[i]) nt++;
        if (pierce[i]) nt++;
    }
    out->n = nt;
    out->p = (float(*)[3])malloc(sizeof(float[3]) * (size_t)(nt > 0 ? nt : 1));
    int j = 0;
    for (int i = 0; i < n; i++) {
        if (inside[i]) { memcpy(out->p[j], in->p[i], sizeof(float[3])); j++; }
        if (pierce[i]) { memcpy(out->p[j], pp[i], sizeof(float[3])); j++; }
    }
    free(pp); free(pierce);
}

/* geometry.f90:239-256 */
static void trim_polygon_more(const polygon *in, const halfspace *hs, int nh, polygon *out)
{
    polygon temp = { in->n, (float(*)[3])malloc(sizeof(float[3]) * (size_t)(in->n > 0 ? in->n : 1)) };
    memcpy(temp.p, in->p, sizeof(float[3]) * (size_t)in->n);
    out->n = 0; out->p = NULL;
    for (int ic = 0; ic < nh; ic++) {
        if (ic != 0) { free(temp.p); temp = *out; }
        trim_polygon_one(&temp, &hs[ic], out);
    }
    free(temp.p);
}

/* exported for the reference's test_geometry.f90 vectors (tests/test_oracle_kats.py) */
int ko_point_in_halfspace(const float pt[3], const float hpoint[3], const float hnormal[3])
{
    halfspace h;
    memcpy(h.point, hpoint, sizeof(h.point)); memcpy(h.normal, hnormal, sizeof(h.normal));
    return point_in_halfspace(pt, &h);
}

void ko_get_piercingpoint(const float a[3], const float b[3], const float hpoint[3], const float hnormal[3], float pp[3],
                          int *between_ab, int *parallel)
{
    halfspace h;
    memcpy(h.point, hpoint, sizeof(h.point)); memcpy(h.normal, hnormal, sizeof(h.normal));
    get_piercingpoint(a, b, &h, pp, between_ab, parallel, NULL, NULL);
}

/* circle_to_polygon + trim_polygon with one half-space; returns the number of points written to out[][3] */
int ko_trim_circle(const float center[3], const float transform[9], int npoints, const float hpoint[3], const float hnormal[3],
                   float *out, int maxn)
{
    float tr[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tr[i][j] = transform[3 * i + j];
    halfspace h;
    memcpy(h.point, hpoint, sizeof(h.point)); memcpy(h.normal, hnormal, sizeof(h.normal));
    polygon circ, trimmed;
    circle_to_polygon(center, tr, npoints, &circ);
    trim_polygon_one(&circ, &h, &trimmed);
    const int n = trimmed.n < maxn ? trimmed.n : maxn;
    memcpy(out, trimmed.p, sizeof(float[3]) * (size_t)n);
    free(circ.p); free(trimmed.p);
    return trimmed.n;
}

/* ---------------------------------------------------------------- heap.f90 (1-based indices kept) */
typedef struct { int *iheap; int n, cap; } iheap;

static void swapi(int *a, int *b) { int t = *a; *a = *b; *b = t; }

static void upheap(iheap *h, int element, const float *keys, int *bp)       /* :205-229 */
{
    int v = element;
    while (v > 1) {
        int u = (v - 2) / 2 + 1;
        if (keys[h->iheap[u] - 1] <= keys[h->iheap[v] - 1]) return;
        swapi(&h->iheap[u], &h->iheap[v]);
        swapi(&bp[h->iheap[u] - 1], &bp[h->iheap[v] - 1]);
        v = u;
    }
}

static void downheap(iheap *h, int element, const float *keys, int *bp)     /* :172-203 */
{
    int v = element, w = 2 * (v - 1) + 2;
    while (w <= h->n) {
        if (w + 1 <= h->n && keys[h->iheap[w + 1] - 1] < keys[h->iheap[w] - 1]) w = w + 1;
        if (keys[h->iheap[v] - 1] <= keys[h->iheap[w] - 1]) return;
        swapi(&h->iheap[v], &h->iheap[w]);
        swapi(&bp[h->iheap[v] - 1], &bp[h->iheap[w] - 1]);
        v = w;
        w = 2 * (v - 1) + 2;
    }
}

static void pushheap(iheap *h, int keyindex, const float *keys, int *bp)    /* :76-101 */
{
    if (h->n + 1 > h->cap) return;
    h->n++;
    h->iheap[h->n] = keyindex;
    bp[keyindex - 1] = h->n;
    upheap(h, h->n, keys, bp);
}

static int popheap(iheap *h, const float *keys, int *bp)                    /* :103-131 */
{
    if (h->n == 0) return 0;
    swapi(&h->iheap[1], &h->iheap[h->n]);
    swapi(&bp[h->iheap[1] - 1], &bp[h->iheap[h->n] - 1]);
    bp[h->iheap[h->n] - 1] = 0;
    int keyindex = h->iheap[h->n];
    h->n--;
    downheap(h, 1, keys, bp);
    return keyindex;
}

static void updateheap(iheap *h, int keyindex, float newkey, float *keys, int *bp)   /* :133-156 */
{
    float oldkey = keys[keyindex - 1];
    keys[keyindex - 1] = newkey;
    if (newkey < oldkey) upheap(h, bp[keyindex - 1], keys, bp);
    if (newkey > oldkey) downheap(h, bp[keyindex - 1], keys, bp);
}

/* ---------------------------------------------------------------- eikonal.f90:29-199
 * arrays are (nx, ny) column-major: element (ix,iy) 1-based at [(iy-1)*nx + ix - 1] */
void ko_eikonal_solver_fmm(const float *speed, int nx, int ny, const float origin[2], const float delta[2],
                           const float initialpoint[2], float *times)
{
    const int FARAWAY = -1, ALIVE = 0;
    const float infinity = 3.40282347e+38f * 0.1f;
    const float dx = delta[0], dy = delta[1];
    int *bp = (int *)malloc(sizeof(int) * (size_t)(nx * ny));
    iheap heap = { (int *)malloc(sizeof(int) * (size_t)(nx * ny + 2)), 0, nx * ny };
    for (int i = 0; i < nx * ny; i++) bp[i] = FARAWAY;
    int ix = (int)((initialpoint[0] - origin[0]) / dx) + 1;
    int iy = (int)((initialpoint[1] - origin[1]) / dy) + 1;
    if (ix < 1) ix = 1;
    if (nx < ix) ix = nx;
    if (iy < 1) iy = 1;
    if (ny < iy) iy = ny;
    for (int i = 0; i < nx * ny; i++) times[i] = infinity;
#define IND(x, y) (((y) - 1) * nx + (x))
#define T(x, y) times[IND(x, y) - 1]
#define S(x, y) speed[IND(x, y) - 1]
    T(ix, iy) = 0.0f;
    if (nx == 1 && ny == 1) { free(bp); free(heap.iheap); return; }
    bp[IND(ix, iy) - 1] = ALIVE;
    int nalive = 1;
    if (1 < ix) T(ix - 1, iy) = dx / S(ix - 1, iy);
    if (ix < nx) T(ix + 1, iy) = dx / S(ix + 1, iy);
    if (1 < iy) T(ix, iy - 1) = dy / S(ix, iy - 1);
    if (iy < ny) T(ix, iy + 1) = dy / S(ix, iy + 1);
    if (1 < ix) pushheap(&heap, IND(ix - 1, iy), times, bp);
    if (ix < nx) pushheap(&heap, IND(ix + 1, iy), times, bp);
    if (1 < iy) pushheap(&heap, IND(ix, iy - 1), times, bp);
    if (iy < ny) pushheap(&heap, IND(ix, iy + 1), times, bp);
    while (nalive <= nx * ny) {
        int imin = popheap(&heap, times, bp);
        if (imin == 0) break;
        ix = (imin - 1) % nx + 1;
        iy = (imin - 1) / nx + 1;
        bp[imin - 1] = ALIVE;
        nalive++;
        const int nbx[4] = { ix - 1, ix + 1, ix, ix }, nby[4] = { iy, iy, iy - 1, iy + 1 };
        const int ok[4] = { 1 < ix, ix < nx, 1 < iy, iy < ny };
        for (int q = 0; q < 4; q++) {
            if (!ok[q]) continue;
            const int x = nbx[q], y = nby[q];        /* update_neighbor :121-186 */
            const int i = IND(x, y);
            if (bp[i - 1] == ALIVE) continue;
            if (bp[i - 1] == FARAWAY) pushheap(&heap, i, times, bp);
            float a = infinity, b = infinity, c = infinity, d = infinity;
            const float told = T(x, y);
            if (1 < x) a = T(x - 1, y);
            if (x < nx) b = T(x + 1, y);
            if (1 < y) c = T(x, y - 1);
            if (y < ny) d = T(x, y + 1);
            float t = 0.f;
            const float aa = fminf(a, b), cc = fminf(c, d);
            const float sp = S(x, y);
            if (fmaxf(aa, cc) != infinity) {
                const float s = (dx * dx) * (dy * dy) * ((dx * dx) + (dy * dy) - ((aa - cc) * sp) * ((aa - cc) * sp));
                if (s >= 0.f)
                    t = fmaxf(t, ((aa * (dy * dy) + cc * (dx * dx)) * sp + sqrtf(s)) / (sp * ((dx * dx) + (dy * dy))));
            }
            if (fminf(c, d) == infinity) {
                if (a < infinity) t = fmaxf(t, a + dx / sp);
                if (b < infinity) t = fmaxf(t, b + dx / sp);
            }
            if (fminf(a, b) == infinity) {
                if (c < infinity) t = fmaxf(t, c + dy / sp);
                if (d < infinity) t = fmaxf(t, d + dy / sp);
            }
            if (t == 0.f) {
                t = infinity;
                if (a < infinity) t = fminf(t, a + dx / sp);
                if (b < infinity) t = fminf(t, b + dx / sp);
                if (c < infinity) t = fminf(t, c + dy / sp);
                if (d < infinity) t = fminf(t, d + dy / sp);
            }
            if (t != 0.f && told != t) updateheap(&heap, i, t, times, bp);
        }
    }
#undef IND
#undef T
#undef S
    free(bp); free(heap.iheap);
}

/* ---------------------------------------------------------------- source_(mt_)eikonal.f90 */
typedef struct {
    int mt;                 /* 1: mt_eikonal (20 params), 0: eikonal (15 params) */
    const float *P;         /* wire-order parameters */
    float rot_rup[3][3], rot_slip[3][3];
    int ncon;
    halfspace con[8];
    /* positions of the shared parameters (0-based) */
    int i_bsx, i_bsy, i_brad, i_nx, i_ny, i_relv;
} eik;

static void rc_to_ned(const eik *e, const float rc[3], float o[3])      /* source_mt_eikonal.f90:625-630 */
{
    matvec((float(*)[3])e->rot_rup, rc, o);
    for (int k = 0; k < 3; k++) o[k] = o[k] + e->P[1 + k];
}
static void ned_to_rc(const eik *e, const float pt[3], float o[3])      /* :618-623 */
{
    float d[3] = { pt[0] - e->P[1], pt[1] - e->P[2], pt[2] - e->P[3] };
    matTvec((float(*)[3])e->rot_rup, d, o);
}
static int in_constraints(const eik *e, const float pt[3])             /* parameterized_source.f90:170-183 */
{
    for (int i = 0; i < e->ncon; i++) if (!point_in_halfspace(pt, &e->con[i])) return 0;
    return 1;
}

/* discretize_subfault_time, source_mt_eikonal.f90:712-762 */
static int subfault_time(float dursf, float risetime, float maxdt, float **tw, float **to, int *cap)
{
    float durfull = dursf + risetime;
    int nt = (int)floorf(durfull / maxdt) + 1;
    if (nt > *cap) { *cap = nt; *tw = (float *)realloc(*tw, sizeof(float) * (size_t)nt); *to = (float *)realloc(*to, sizeof(float) * (size_t)nt); }
    if (nt == 1) { (*tw)[0] = 1.f; (*to)[0] = 0.f; return 1; }
    ko_plf stf; stf.n = 4;
    if (risetime < dursf) {
        stf.x[0] = (-dursf - risetime) / 2.f; stf.y[0] = 0.f; stf.x[1] = (-dursf + risetime) / 2.f; stf.y[1] = 1.f / dursf;
        stf.x[2] = (dursf - risetime) / 2.f; stf.y[2] = 1.f / dursf; stf.x[3] = (dursf + risetime) / 2.f; stf.y[3] = 0.f;
    } else {
        stf.x[0] = (-risetime - dursf) / 2.f; stf.y[0] = 0.f; stf.x[1] = (-risetime + dursf) / 2.f; stf.y[1] = 1.f / risetime;
        stf.x[2] = (risetime - dursf) / 2.f; stf.y[2] = 1.f / risetime; stf.x[3] = (risetime + dursf) / 2.f; stf.y[3] = 0.f;
    }
    float tbeg = stf.x[0], dt = durfull / (float)nt;
    for (int it = 1; it <= nt; it++)
        ko_plf_integrate_and_centroid(&stf, tbeg + dt * (float)(it - 1), tbeg + dt * (float)it, &(*tw)[it - 1], &(*to)[it - 1]);
    return nt;
}

/* psm_set_* + psm_to_tdsm_* of source_eikonal.f90:205-316,435-710 / source_mt_eikonal.f90:200-323,442-710.
 * sourcetype 4 = eikonal, 5 = mt_eikonal.  Returns ncentroids or -1 (empty rupture area) / -2 (nucleation
 * point outside of the rupture region). */
int ko_psm_to_tdsm_eikonal(int sourcetype, const float *params, float shortest_doi,
                           const ko_crust_profile *prof_speed, int ncon, const float *con_points,
                           const float *con_normals, ko_centroid **out, float *moment, float *risetime,
                           int grid_size[2])
{
    eik e;
    memset(&e, 0, sizeof(e));
    e.mt = (sourcetype == 5);
    e.P = params;
    const int o = e.mt ? 0 : 1;              /* eikonal has slip-rake at position 8 */
    e.i_bsx = 7 + o; e.i_bsy = 8 + o; e.i_brad = 9 + o; e.i_nx = 10 + o; e.i_ny = 11 + o; e.i_relv = 12 + o;
    *moment = params[4];
    *risetime = e.mt ? params[19] : params[14];
    {
        float strike = ko_d2r_r(params[5]), dip = ko_d2r_r(params[6]);
        ko_init_euler(dip, strike, 0.f, e.rot_rup);
        if (!e.mt) ko_init_euler(dip, strike, -ko_d2r_r(params[7]), e.rot_slip);
    }
    e.ncon = ncon;
    for (int i = 0; i < ncon; i++)
        for (int k = 0; k < 3; k++) { e.con[i].point[k] = con_points[3 * i + k]; e.con[i].normal[k] = con_normals[3 * i + k]; }
    const float bsx = params[e.i_bsx], bsy = params[e.i_bsy], brad = params[e.i_brad];

    /* ---- psm_borderline_*: circle -> 180-gon, clipped by the constraints */
    float center[3];
    { float rc[3] = { bsx, bsy, 0.f }; rc_to_ned(&e, rc, center); }
    float transform[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) transform[i][j] = -e.rot_rup[i][j] * brad;
    int ninit = 180;
    if (brad == 0.f) ninit = 1;
    polygon circ, rup;
    circle_to_polygon(center, transform, ninit, &circ);
    trim_polygon_more(&circ, e.con, e.ncon, &rup);
    free(circ.p);
    if (rup.n == 0) { free(rup.p); return -1; }
    float mn[3] = { 3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f }, mx[3] = { -3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f };
    for (int i = 0; i < rup.n; i++) {
        float rc[3];
        ned_to_rc(&e, rup.p[i], rc);
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], rc[k]); mx[k] = fmaxf(mx[k], rc[k]); }
    }
    free(rup.p);

    /* ---- psm_make_*_grid */
    const float deltagrid = fminf(100.f * shortest_doi / 2.f, 4000.f);
    float first[2] = { mn[0], mn[1] }, last[2] = { mx[0], mx[1] };
    float dims[2] = { last[0] - first[0], last[1] - first[1] };
    int nd[2] = { (int)ceilf(dims[0] / deltagrid), (int)ceilf(dims[1] / deltagrid) };
    if (nd[0] == 0) nd[0] = 1;
    if (nd[1] == 0) nd[1] = 1;
    float delta[2] = { dims[0] / (float)nd[0], dims[1] / (float)nd[1] };
    /* psm_initial_point_intolerant_rc */
    float init_rc[3] = { params[e.i_nx], params[e.i_ny], 0.f };
    {
        float nukl = sqrtf(init_rc[0] * init_rc[0] + init_rc[1] * init_rc[1]);
        float ned[3];
        rc_to_ned(&e, init_rc, ned);
        if (!in_constraints(&e, ned) || nukl > brad) return -2;
    }
    const int fx = nd[0], fy = nd[1];
    float *speed = (float *)malloc(sizeof(float) * (size_t)(fx * fy) * 5), *ftimes = speed + fx * fy, *fpts = ftimes + fx * fy;
    float minspeed = 3.40282347e+38f;
    for (int iy = 1; iy <= fy; iy++)
        for (int ix = 1; ix <= fx; ix++) {
            float rc[3] = { first[0] + ((float)ix - 0.5f) * delta[0], first[1] + ((float)iy - 0.5f) * delta[1], 0.f }, pt[3];
            rc_to_ned(&e, rc, pt);
            const int id = (iy - 1) * fx + ix - 1;
            fpts[3 * id] = pt[0]; fpts[3 * id + 1] = pt[1]; fpts[3 * id + 2] = pt[2];
            float dd[3] = { pt[0] - center[0], pt[1] - center[1], pt[2] - center[2] };
            if (sqrtf(dot3(dd, dd)) > brad || !in_constraints(&e, pt)) {
                speed[id] = 0.f;
            } else {
                float vp, vs, rho;
                crust_at_depth(prof_speed, pt[2], &vp, &vs, &rho);
                speed[id] = vs * params[e.i_relv];
                minspeed = fminf(speed[id], minspeed);
            }
        }
    const float invalid_speed = minspeed * 0.5f;
    for (int i = 0; i < fx * fy; i++) if (speed[i] == 0.f) speed[i] = invalid_speed;
    float ip2[2] = { init_rc[0], init_rc[1] };
    ko_eikonal_solver_fmm(speed, fx, fy, first, delta, ip2, ftimes);
    for (int i = 0; i < fx * fy; i++) if (speed[i] == invalid_speed) ftimes[i] = -1.f;

    /* ---- optimal coarse grid, psm_to_tdsm_size_* */
    const float maxdt = shortest_doi, maxdx = 0.5f * shortest_doi * minspeed, maxdy = maxdx;
    const float sizex = last[0] - first[0], sizey = last[1] - first[1];
    int nxc = (int)floorf(sizex / maxdx) + 1;
    if (nxc <= 1) nxc = 2;
    if (sizex == 0.f) nxc = 1;
    int nyc = (int)floorf(sizey / maxdy) + 1;
    if (nyc <= 1) nyc = 2;
    if (sizey == 0.f) nyc = 1;

    /* ---- psm_downsample_grid */
    float cdelta[2] = { (last[0] - first[0]) / (float)nxc, (last[1] - first[1]) / (float)nyc };
    if (cdelta[0] == 0.f || nxc == 0) cdelta[0] = 1.f;
    if (cdelta[1] == 0.f || nyc == 0) cdelta[1] = 1.f;
    const int nc = nxc * nyc;
    float *ntimes = (float *)calloc((size_t)nc * 8, sizeof(float));
    float *ctimes = ntimes + nc, *cspeed = ctimes + nc, *cdur = cspeed + nc, *cw = cdur + nc, *cpts = cw + nc;
    for (int i = 0; i < nc; i++) ctimes[i] = -1.f;
    int npf = 0;
    for (int iyf = 1; iyf <= fy; iyf++)
        for (int ixf = 1; ixf <= fx; ixf++) {
            const int id = (iyf - 1) * fx + ixf - 1;
            if (ftimes[id] < 0.f) continue;
            float rc[3];
            ned_to_rc(&e, &fpts[3 * id], rc);
            const int ixc = (int)floorf((rc[0] - first[0]) / cdelta[0]) + 1, iyc = (int)floorf((rc[1] - first[1]) / cdelta[1]) + 1;
            if (ixc < 1 || iyc < 1 || ixc > nxc || iyc > nyc) continue;      /* 'orphaned point' */
            const int ic = (iyc - 1) * nxc + ixc - 1;
            ntimes[ic] = ntimes[ic] + 1.f;
            if (ctimes[ic] == -1.f) ctimes[ic] = 0.f;
            ctimes[ic] = ctimes[ic] + ftimes[id];
            cspeed[ic] = cspeed[ic] + 1.f / speed[id];
            for (int k = 0; k < 3; k++) cpts[3 * ic + k] = cpts[3 * ic + k] + fpts[3 * id + k];
            npf++;
        }
    for (int ic = 0; ic < nc; ic++) if (ntimes[ic] > 0.f) {
        ctimes[ic] = 1.f / ntimes[ic] * ctimes[ic];
        cspeed[ic] = 1.f / (1.f / ntimes[ic] * cspeed[ic]);
        for (int k = 0; k < 3; k++) cpts[3 * ic + k] = 1.f / ntimes[ic] * cpts[3 * ic + k];
    }
    for (int ic = 0; ic < nc; ic++) cw[ic] = ntimes[ic] / (float)npf;
    for (int iyf = 1; iyf <= fy; iyf++)
        for (int ixf = 1; ixf <= fx; ixf++) {
            const int id = (iyf - 1) * fx + ixf - 1;
            if (ftimes[id] < 0.f) continue;
            float rc[3];
            ned_to_rc(&e, &fpts[3 * id], rc);
            const int ixc = (int)floorf((rc[0] - first[0]) / cdelta[0]) + 1, iyc = (int)floorf((rc[1] - first[1]) / cdelta[1]) + 1;
            if (ixc < 1 || iyc < 1 || ixc > nxc || iyc > nyc) continue;
            const int ic = (iyc - 1) * nxc + ixc - 1;
            cdur[ic] = cdur[ic] + fabsf(ftimes[id] - ctimes[ic]);
        }
    for (int ic = 0; ic < nc; ic++) if (ntimes[ic] > 0.f) cdur[ic] = 4.f / ntimes[ic] * cdur[ic];

    /* ---- psm_to_tdsm_table_* */
    const float origin_time = params[0];
    int ndc = 0;
    float centertime = 0.f;
    for (int iy = 1; iy <= nyc; iy++)
        for (int ix = 1; ix <= nxc; ix++) {
            const int ic = (iy - 1) * nxc + ix - 1;
            if (ctimes[ic] >= 0.f) {
                ndc += (int)floorf((cdur[ic] + 0.0f) / maxdt) + 1;
                centertime = centertime + ctimes[ic] * cw[ic];
            }
        }
    float m6[6];
    if (e.mt) {
        for (int k = 0; k < 6; k++) m6[k] = params[13 + k];
    } else {       /* m_rot = R m_unrot R^T, source_eikonal.f90:676-681 */
        float mu[3][3] = { { 0, 0, -1 }, { 0, 0, 0 }, { -1, 0, 0 } }, tr[3][3], inner[3][3], mr[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tr[i][j] = e.rot_slip[j][i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inner[i][j] = (mu[i][0] * tr[0][j] + mu[i][1] * tr[1][j]) + mu[i][2] * tr[2][j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mr[i][j] = (e.rot_slip[i][0] * inner[0][j] + e.rot_slip[i][1] * inner[1][j]) + e.rot_slip[i][2] * inner[2][j];
        m6[0] = mr[0][0]; m6[1] = mr[1][1]; m6[2] = mr[2][2]; m6[3] = mr[0][1]; m6[4] = mr[0][2]; m6[5] = mr[1][2];
    }
    ko_centroid *c = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)(ndc > 0 ? ndc : 1));
    float *tw = NULL, *to = NULL;
    int cap = 0, id = 0;
    for (int iy = 1; iy <= nyc; iy++)
        for (int ix = 1; ix <= nxc; ix++) {
            const int ic = (iy - 1) * nxc + ix - 1;
            if (ctimes[ic] < 0.f) continue;
            const int nt = subfault_time(cdur[ic], 0.f, maxdt, &tw, &to, &cap);
            for (int it = 0; it < nt; it++) {
                c[id].north = cpts[3 * ic]; c[id].east = cpts[3 * ic + 1]; c[id].depth = cpts[3 * ic + 2];
                c[id].time = ctimes[ic] + to[it] + origin_time - centertime;
                for (int k = 0; k < 6; k++) c[id].m[k] = m6[k] * tw[it] * cw[ic];
                id++;
            }
        }
    free(tw); free(to); free(ntimes); free(speed);
    grid_size[0] = nxc; grid_size[1] = nyc;
    *out = c;
    return id;
}
