/*
 * ko_gfdb.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates the READ side of gfdb.f90 over an in-memory trace table: index
 * maths (gfdb.f90:781-815), trace fetch (gfdb.f90:830-863) and the bilinear
 * 4-neighbour blend (gfdb.f90:865-950).  The chunked HDF5 cache / LRU
 * (gfdb.f90:952-1031, gfdb_io_hdf.f90) is I/O only and is replaced by a flat
 * array of gap-compressed traces.
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <omp.h>

ko_gfdb *ko_gfdb_create(int nx, int nz, int ng, float dt, float dx, float dz, float firstx, float firstz)
{
    ko_gfdb *db = (ko_gfdb *)calloc(1, sizeof(ko_gfdb));
    db->nx = nx; db->nz = nz; db->ng = ng;
    db->dt = dt; db->dx = dx; db->dz = dz; db->firstx = firstx; db->firstz = firstz;
    db->traces = (ko_trace *)calloc((size_t)nx * nz * ng, sizeof(ko_trace));
    db->nscratch = omp_get_max_threads() > 256 ? omp_get_max_threads() : 256;
    db->scratch = (ko_trace *)calloc((size_t)db->nscratch, sizeof(ko_trace));
    return db;
}

void ko_gfdb_destroy(ko_gfdb *db)
{
    if (!db) return;
    for (size_t i = 0; i < (size_t)db->nx * db->nz * db->ng; i++) ko_trace_destroy(&db->traces[i]);
    for (int i = 0; i < db->nscratch; i++) ko_trace_destroy(&db->scratch[i]);
    free(db->traces); free(db->scratch); free(db);
}

static ko_trace *slot(const ko_gfdb *db, int ix, int iz, int ig)   /* 1-based */
{
    return &db->traces[((size_t)(ix - 1) * db->nz + (iz - 1)) * db->ng + (ig - 1)];
}

/* as gfdb_build does: a dense strip is trace_pack'ed (sparse_trace.f90:443) before gfdb_save_trace */
void ko_gfdb_set_trace_dense(ko_gfdb *db, int ix, int iz, int ig, int lo, int hi, const float *data)
{
    ko_strip s = { NULL, 1, 0 };
    ko_strip_init(&s, lo, hi, data);
    ko_trace_pack(&s, slot(db, ix, iz, ig));
    ko_strip_destroy(&s);
}

int ko_gfdb_trace_span(const ko_gfdb *db, int ix, int iz, int ig, int span[2])
{
    if (ix < 1 || ix > db->nx || iz < 1 || iz > db->nz || ig < 1 || ig > db->ng) return 0;
    const ko_trace *t = slot(db, ix, iz, ig);
    if (!t->strips) return 0;
    span[0] = t->span[0]; span[1] = t->span[1];
    return 1;
}

void ko_gfdb_trace_unpack(const ko_gfdb *db, int ix, int iz, int ig, float *out)
{
    const ko_trace *t = slot(db, ix, iz, ig);
    ko_strip s = { NULL, 1, 0 };
    ko_trace_unpack(t, &s);
    for (int i = 0; i < s.n; i++) out[i] = s.d[i];
    ko_strip_destroy(&s);
}

/* Fortran nint(): round half away from zero */
static int nint_f(float x) { return (int)roundf(x); }

/* gfdb.f90:781-792 */
void ko_gfdb_get_indices(const ko_gfdb *c, float x, float z, int *ix, int *iz)
{
    *ix = nint_f((x - c->firstx) / c->dx) + 1;
    *iz = nint_f((z - c->firstz) / c->dz) + 1;
}

/* gfdb.f90:794-815 */
void ko_gfdb_get_indices_bilin(const ko_gfdb *c, float x, float z, int xus, int zus,
                               int ix[2], int iz[2], float *dix, float *diz)
{
    ix[0] = (int)floorf((x - c->firstx) / (c->dx * (float)xus)) * xus + 1;
    iz[0] = (int)floorf((z - c->firstz) / (c->dz * (float)zus)) * zus + 1;
    ix[1] = ix[0] + xus;
    iz[1] = iz[0] + zus;
    *dix = (x - c->firstx - (float)(ix[0] - 1) * c->dx) / (c->dx * (float)xus);
    *diz = (z - c->firstz - (float)(iz[0] - 1) * c->dz) / (c->dz * (float)zus);
}

/* gfdb.f90:830-863 + chunk_get_trace's "no trace available" (gfdb.f90:1003-1008): NULL when
 * out of bounds or not stored */
static const ko_trace *get_trace(const ko_gfdb *db, int ix, int iz, int ig)
{
    if (ix > db->nx || ix < 1 || iz > db->nz || iz < 1 || ig > db->ng || ig < 1) return NULL;
    const ko_trace *t = slot(db, ix, iz, ig);
    if (!t->strips) return NULL;
    return t;
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* gfdb.f90:865-950.  Returns a borrowed trace: either a stored one or this
 * thread's blend buffer, valid until the thread's next bilinear get. */
const ko_trace *ko_gfdb_get_trace_bilin(ko_gfdb *db, const int ix[2], const int iz[2], int ig,
                                        float dix, float diz)
{
    if (dix == 0.f && diz == 0.f) return get_trace(db, ix[0], iz[0], ig);
    const ko_trace *t00 = get_trace(db, ix[0], iz[0], ig);
    const ko_trace *t01 = get_trace(db, ix[0], iz[1], ig);
    const ko_trace *t10 = get_trace(db, ix[1], iz[0], ig);
    const ko_trace *t11 = get_trace(db, ix[1], iz[1], ig);
    if (!(t00 && t01 && t10 && t11)) return NULL;
    int span[2];
    span[0] = imin(imin(t00->span[0], t01->span[0]), imin(t10->span[0], t11->span[0]));
    span[1] = imax(imax(t00->span[1], t01->span[1]), imax(t10->span[1], t11->span[1]));
    ko_trace *tp = &db->scratch[omp_get_thread_num()];
    int n = span[1] - span[0] + 1;
    if (!tp->strips || tp->strips[0].n != n || tp->strips[0].lo != span[0]) {
        float *z = (float *)calloc((size_t)n, sizeof(float));
        ko_trace_create_simple(tp, z, span[0], span[1]);
        free(z);
    }
    tp->span[0] = span[0]; tp->span[1] = span[1];
    float *a = tp->strips[0].d;
    for (int i = 0; i < n; i++) a[i] = 0.f;
    ko_trace_multiply_add_nogrow(t00, a, span[0], span[1], (1.f - dix) * (1.f - diz), 0, 0, 0.f);
    ko_trace_multiply_add_nogrow(t01, a, span[0], span[1], (1.f - dix) * diz, 0, 0, 0.f);
    ko_trace_multiply_add_nogrow(t10, a, span[0], span[1], dix * (1.f - diz), 0, 0, 0.f);
    ko_trace_multiply_add_nogrow(t11, a, span[0], span[1], dix * diz, 0, 0, 0.f);
    return tp;
}
