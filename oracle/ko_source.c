/*
 * ko_source.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates the source discretisers: parameterised source -> centroid table
 *   source_moment_tensor.f90:163-267, source_bilat.f90:173-459,
 *   source_circular.f90:165-444 (psm_set_* and psm_to_tdsm_*).
 * All arithmetic is default real (fp32) as in the reference.
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int ko_psm_nparams(int sourcetype)
{
    switch (sourcetype) {
    case KO_SRC_BILAT: return 14;           /* source_bilat.f90:32 */
    case KO_SRC_CIRCULAR: return 11;        /* source_circular.f90:32 */
    case KO_SRC_MOMENT_TENSOR: return 11;   /* source_moment_tensor.f90:34 */
    case KO_SRC_POINT_LP: return 13;        /* source_point_lp.f90:43 */
    }
    return -1;
}

/* 3x3 matmul / matvec, summed k = 1,2,3 */
static void matmul33(float a[3][3], float b[3][3], float c[3][3])
{
    float t[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[i][j] = (a[i][0] * b[0][j] + a[i][1] * b[1][j]) + a[i][2] * b[2][j];
    memcpy(c, t, sizeof(t));
}
static void matvec3(float a[3][3], const float v[3], float p[3])
{
    for (int i = 0; i < 3; i++)
        p[i] = (a[i][0] * v[0] + a[i][1] * v[1]) + a[i][2] * v[2];
}

/* source_bilat.f90:216-239 / source_circular.f90:209-231 (rotation matrices only).
 * NB: source_circular.f90:221 takes params(9) (= radius) as 'rupdir'; kept. */
static void update_dep_params(ko_psm *psm)
{
    float strike = ko_d2r_r(psm->params[5]);
    float dip = ko_d2r_r(psm->params[6]);
    float rake = ko_d2r_r(psm->params[7]);
    float rupdir = ko_d2r_r(psm->params[8]);
    ko_init_euler(dip, strike, -rupdir, psm->rotmat_rup);
    ko_init_euler(dip, strike, -rake, psm->rotmat_slip);
}

int ko_psm_set(ko_psm *psm, int sourcetype, const float *params)
{
    int n = ko_psm_nparams(sourcetype);
    if (n < 0) return -1;
    int only_moment_changed = 0;
    if (sourcetype == KO_SRC_BILAT || sourcetype == KO_SRC_CIRCULAR || sourcetype == KO_SRC_POINT_LP) {
        /* source_bilat.f90:206, source_circular.f90:199, source_point_lp.f90:226 */
        if (psm->inited && psm->sourcetype == sourcetype) {
            int cnt = 0;
            for (int i = 0; i < n; i++) if (params[i] != psm->params[i]) cnt++;
            only_moment_changed = (cnt <= 1 && params[4] != psm->params[4]);
        }
    }
    if (!psm->inited || psm->sourcetype != sourcetype) {
        psm->grid_size[0] = psm->grid_size[1] = psm->grid_size[2] = 1;
    }
    psm->sourcetype = sourcetype;
    psm->nparams = n;
    memcpy(psm->params, params, sizeof(float) * (size_t)n);
    psm->risetime = 0.f;                                   /* parameterized_source.f90:71 */
    if (sourcetype == KO_SRC_MOMENT_TENSOR) {
        psm->moment = 1.f;                                 /* source_moment_tensor.f90:201 */
    } else if (sourcetype == KO_SRC_POINT_LP) {
        psm->moment = psm->params[4];                      /* source_point_lp.f90:230 */
    } else {
        psm->moment = psm->params[4];                      /* source_bilat.f90:210 */
        update_dep_params(psm);
    }
    psm->inited = 1;
    return only_moment_changed;
}

static void plf4(ko_plf *s, float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4)
{
    s->n = 4;
    s->x[0] = x1; s->y[0] = y1; s->x[1] = x2; s->y[1] = y2;
    s->x[2] = x3; s->y[2] = y3; s->x[3] = x4; s->y[3] = y4;
}

/* trapezoid STF = box(risetime) * box(dursf), binned into nt weights/offsets
 * source_bilat.f90:386-416, source_circular.f90:370-400 */
static void stf_bins(float dursf, float risetime, int nt, float *wt, float *toff)
{
    ko_plf stf;
    if (risetime < dursf)
        plf4(&stf, (-dursf - risetime) / 2.f, 0.f, (-dursf + risetime) / 2.f, 1.f / dursf,
                   (dursf - risetime) / 2.f, 1.f / dursf, (dursf + risetime) / 2.f, 0.f);
    else
        plf4(&stf, (-risetime - dursf) / 2.f, 0.f, (-risetime + dursf) / 2.f, 1.f / risetime,
                   (risetime - dursf) / 2.f, 1.f / risetime, (risetime + dursf) / 2.f, 0.f);
    float durfull = dursf + risetime;
    float tbeg = stf.x[0];
    float dt = durfull / (float)nt;
    for (int it = 1; it <= nt; it++) {
        float ta = tbeg + dt * (float)(it - 1);
        float tb = tbeg + dt * (float)it;
        ko_plf_integrate_and_centroid(&stf, ta, tb, &wt[it - 1], &toff[it - 1]);
    }
}

/* rotated double couple, divided by the number of sub-faults
 * source_bilat.f90:424-438, source_circular.f90:409-416 */
static void rotated_mt(ko_psm *psm, int np, float m_rot[3][3])
{
    float m_unrot[3][3] = { { 0, 0, -1 }, { 0, 0, 0 }, { -1, 0, 0 } };
    float trot[3][3], inner[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) trot[i][j] = psm->rotmat_slip[j][i];
    matmul33(m_unrot, trot, inner);
    matmul33(psm->rotmat_slip, inner, m_rot);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m_rot[i][j] = m_rot[i][j] / (float)np;
}

static void fill_centroids(ko_centroid *c, int np, int nt, const float *grid /*[np][3]*/,
                           const float *tshift, const float *wt, const float *toff, float m_rot[3][3])
{
    int id = 0;
    for (int ip = 0; ip < np; ip++)
        for (int it = 0; it < nt; it++) {
            c[id].north = grid[3 * ip + 0];
            c[id].east = grid[3 * ip + 1];
            c[id].depth = grid[3 * ip + 2];
            c[id].time = tshift[ip] + toff[it];
            c[id].m[0] = m_rot[0][0] * wt[it];
            c[id].m[1] = m_rot[1][1] * wt[it];
            c[id].m[2] = m_rot[2][2] * wt[it];
            c[id].m[3] = m_rot[0][1] * wt[it];
            c[id].m[4] = m_rot[0][2] * wt[it];
            c[id].m[5] = m_rot[1][2] * wt[it];
            id++;
        }
}

/* source_moment_tensor.f90:205-267 */
static int to_tdsm_moment_tensor(ko_psm *psm, float shortest_doi, ko_centroid **out)
{
    const float *p = psm->params;
    float risetime = p[10], time = p[0];
    float maxdt = shortest_doi;
    int nt = (int)floorf(risetime / maxdt) + 1;
    if (nt <= 1) nt = 2;
    psm->grid_size[0] = nt;
    ko_plf stf;
    plf4(&stf, (-risetime) / 2.f, 0.f, (-risetime) / 2.f, 1.f / risetime,
               (risetime) / 2.f, 1.f / risetime, (risetime) / 2.f, 0.f);
    float tbeg = stf.x[0];
    float dt = risetime / (float)nt;
    float *wt = (float *)malloc(sizeof(float) * 2 * (size_t)nt), *toff = wt + nt;
    for (int it = 1; it <= nt; it++) {
        float ta = tbeg + dt * (float)(it - 1);
        float tb = tbeg + dt * (float)it;
        ko_plf_integrate_and_centroid(&stf, ta, tb, &wt[it - 1], &toff[it - 1]);
    }
    ko_centroid *c = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)nt);
    for (int it = 0; it < nt; it++) {
        c[it].north = p[1]; c[it].east = p[2]; c[it].depth = p[3];
        c[it].time = toff[it] + time;
        for (int k = 0; k < 6; k++) c[it].m[k] = p[4 + k] * wt[it];
    }
    free(wt);
    *out = c;
    return nt;
}

/* source_bilat.f90:241-459 */
static int to_tdsm_bilat(ko_psm *psm, float shortest_doi, ko_centroid **out)
{
    const float *P = psm->params;
    float rupvel = P[12];
    float maxdt = shortest_doi;
    float maxdx = 0.5f * shortest_doi * rupvel;
    float maxdy = shortest_doi * rupvel;
    /* psm_to_tdsm_size_bilat :274-315 */
    float length_a = P[9], length_b = P[10], width = P[11], risetime = P[13];
    float length = length_a + length_b;
    int nx = (int)floorf(length / maxdx) + 1;
    if (nx <= 1) nx = 2;
    if (length == 0.f) nx = 1;
    int ny = (int)floorf(width / maxdy) + 1;
    if (ny <= 1) ny = 2;
    if (width == 0.f) ny = 1;
    float dursf = length / (float)nx / rupvel;
    float durfull = risetime + dursf;
    int nt = (int)floorf(durfull / maxdt) + 1;
    if (nt <= 1) nt = 2;
    /* psm_to_tdsm_table_bilat :318-459 */
    float north = P[1], east = P[2], depth = P[3];
    int np = nx * ny;
    float *grid = (float *)malloc(sizeof(float) * (size_t)(4 * np + 2 * nt));
    float *tshift = grid + 3 * np, *wt = tshift + np, *toff = wt + nt;
    int ip = 0;
    for (int ix = 1; ix <= nx; ix++)
        for (int iy = 1; iy <= ny; iy++) {
            float g[3], p[3];
            g[0] = (2.f * ((float)ix - 1.f) - (float)nx + 1.f) / (2.f * (float)nx) * length;
            g[1] = (2.f * ((float)iy - 1.f) - (float)ny + 1.f) / (2.f * (float)ny) * width;
            g[2] = 0.f;
            tshift[ip] = fabsf(length / 2.f - length_b + g[0]) / rupvel + P[0]
                         - fmaxf(length_a, length_b) / 2.f / rupvel;
            matvec3(psm->rotmat_rup, g, p);
            grid[3 * ip + 0] = p[0] + north; grid[3 * ip + 1] = p[1] + east; grid[3 * ip + 2] = p[2] + depth;
            ip++;
        }
    dursf = length / (float)nx / rupvel;
    stf_bins(dursf, risetime, nt, wt, toff);
    float m_rot[3][3];
    rotated_mt(psm, np, m_rot);
    ko_centroid *c = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)(np * nt));
    fill_centroids(c, np, nt, grid, tshift, wt, toff, m_rot);
    free(grid);
    psm->grid_size[0] = nx; psm->grid_size[1] = ny; psm->grid_size[2] = nt;
    *out = c;
    return np * nt;
}

/* source_circular.f90:235-444 */
static int to_tdsm_circular(ko_psm *psm, float shortest_doi, ko_centroid **out)
{
    const float *P = psm->params;
    float rupvel = P[9];
    float maxdt = shortest_doi;
    float maxdx = 0.5f * shortest_doi * rupvel;
    float radius = P[8], risetime = P[10];
    float length = radius * 2.f;
    int nx = (int)floorf(length / maxdx) + 1;
    if (nx <= 1) nx = 2;
    if (length == 0.f) nx = 1;
    int ny = nx;
    float dursf = length / (float)nx / rupvel;
    float durfull = risetime + dursf;
    int nt = (int)floorf(durfull / maxdt) + 1;
    if (nt <= 1) nt = 2;
    float time = P[0], north = P[1], east = P[2], depth = P[3];
    length = 2.f * radius;
    float *grid = (float *)malloc(sizeof(float) * (size_t)(4 * nx * ny + 2 * nt));
    float *tshift = grid + 3 * nx * ny, *wt = tshift + nx * ny, *toff = wt + nt;
    int ip = 0;
    for (int ix = 1; ix <= nx; ix++)
        for (int iy = 1; iy <= ny; iy++) {
            float x = (2.f * ((float)ix - 1.f) - (float)nx + 1.f) / (2.f * (float)nx) * length;
            float y = (2.f * ((float)iy - 1.f) - (float)ny + 1.f) / (2.f * (float)ny) * length;
            float r = sqrtf(x * x + y * y);
            float v[3] = { x, y, 0.f }, p[3];
            matvec3(psm->rotmat_rup, v, p);
            p[0] = p[0] + north; p[1] = p[1] + east; p[2] = p[2] + depth;
            if (r <= radius) {
                grid[3 * ip + 0] = p[0]; grid[3 * ip + 1] = p[1]; grid[3 * ip + 2] = p[2];
                tshift[ip] = r / rupvel + time;
                ip++;
            }
        }
    int np = ip;
    dursf = length / (float)nx / rupvel;
    stf_bins(dursf, risetime, nt, wt, toff);
    float m_rot[3][3];
    rotated_mt(psm, np, m_rot);
    ko_centroid *c = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)(np * nt > 0 ? np * nt : 1));
    fill_centroids(c, np, nt, grid, tshift, wt, toff, m_rot);
    free(grid);
    psm->grid_size[0] = nx; psm->grid_size[1] = ny; psm->grid_size[2] = nt;
    *out = c;
    return np * nt;
}

/* source_all.f90:431-465 dispatch */
/* stf, source_point_lp.f90:408-419: default-real exp / sin, pi from constants.f90 */
static float point_lp_stf(float reltime, float prd, float dur_exc)
{
    const float pi = 3.14159265358979f;
    float t1 = 2.f;
    float t2 = t1 + dur_exc - 5.f;
    float t3 = t2 / 4.f;
    float d = reltime - t3;
    return expf(-(d * d) / (2.f * pi * dur_exc)) * 1.f / (1.f + expf(-2.f * (reltime - t1))) * 1.f /
           (1.f + expf(0.5f * (reltime - t2))) * sinf(2.f * pi / prd * reltime);
}

/* psm_to_tdsm_point_lp + _table_, source_point_lp.f90:237-337 */
static int to_tdsm_point_lp(ko_psm *psm, float shortest_doi, ko_centroid **out)
{
    const float *P = psm->params;
    const float maxdt = shortest_doi, dur_exc = P[11], prd = P[12];
    int nt = (int)floorf(dur_exc / maxdt) + 1;
    if (nt <= 1) nt = 2;
    ko_centroid *c = (ko_centroid *)malloc(sizeof(ko_centroid) * (size_t)nt);
    for (int it = 1; it <= nt; it++) {
        float rel_time = (float)(it - 1) * maxdt;
        float tfactor = point_lp_stf(rel_time, prd, dur_exc);
        c[it - 1].north = P[1]; c[it - 1].east = P[2]; c[it - 1].depth = P[3];
        c[it - 1].time = P[0] + (float)it * maxdt;
        for (int k = 0; k < 6; k++) c[it - 1].m[k] = P[5 + k] * tfactor;
    }
    psm->grid_size[0] = 1; psm->grid_size[1] = 1; psm->grid_size[2] = nt;
    *out = c;
    return nt;
}

int ko_psm_to_tdsm(ko_psm *psm, float shortest_doi, ko_centroid **out)
{
    switch (psm->sourcetype) {
    case KO_SRC_POINT_LP: return to_tdsm_point_lp(psm, shortest_doi, out);
    case KO_SRC_MOMENT_TENSOR: return to_tdsm_moment_tensor(psm, shortest_doi, out);
    case KO_SRC_BILAT: return to_tdsm_bilat(psm, shortest_doi, out);
    case KO_SRC_CIRCULAR: return to_tdsm_circular(psm, shortest_doi, out);
    }
    return -1;
}

/* P and T axes of a bilateral source: psm_update_dep_params_bilat (source_bilat.f90:216-239) with polar, domeshot and wrap
 * (:565-593); (azimuth, polar angle) in degrees, lower hemisphere */
static float wrapf_(float x, float mi, float ma) { return x - floorf((x - mi) / (ma - mi)) * (ma - mi); }

void ko_principal_axes_bilat(const float *params, float pax[2], float tax[2])
{
    const float pi = 3.14159265358979f;
    const float d2r = 2.f / 360.f * pi, r2d = 360.f / 2.f / pi;
    float R[3][3];
    ko_init_euler(d2r * params[6], d2r * params[5], -(d2r * params[7]), R);
    const float s2 = sqrtf(2.f);
    for (int a = 0; a < 2; a++) {
        const float v[3] = { a == 0 ? s2 : -s2, 0.f, -s2 };
        float xyz[3], pol[3];
        for (int i = 0; i < 3; i++) xyz[i] = (R[i][0] * v[0] + R[i][1] * v[1]) + R[i][2] * v[2];
        pol[0] = sqrtf((xyz[0] * xyz[0] + xyz[1] * xyz[1]) + xyz[2] * xyz[2]);
        pol[1] = atan2f(xyz[1], xyz[0]);
        pol[2] = acosf(xyz[2] / pol[0]);
        float d1 = wrapf_(pol[1], pi, -pi), d2 = wrapf_(pol[2], pi, -pi);
        if (d2 > pi / 2.f) { d1 = wrapf_(d1 + pi, -pi, pi); d2 = pi - d2; }
        float *out = a == 0 ? pax : tax;
        out[0] = r2d * d1; out[1] = r2d * d2;
    }
}
