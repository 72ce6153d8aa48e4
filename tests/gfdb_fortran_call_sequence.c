/* gfdb_fortran_call_sequence.c -- TEST INFRASTRUCTURE: a second, independent producer of Kiwi Green's function database files
 * for tests/test_gfdb_hdf5.py.  The product's own writer (kiwi_amd/gfdb/kiwi_gfdb_hdf5.c) writes a chunk in one sweep (all
 * references at once); this program instead issues the HDF5 calls in the order and with the arguments the reference's Fortran
 * layer issues them, one stored trace at a time -- gfdb_io_hdf.f90:
 *     gfdb_io_create_index   :181-234   h5_save_scalar_{real,integer} :683-741 (H5S_SCALAR dataspace, one dataset per scalar)
 *     gfdb_io_create_chunk   :236-312   3-D "index" dataset of H5T_STD_REF_OBJ, Fortran dims (ng, nz, nxc) = C dims (nxc, nz, ng),
 *                                        written as zeros; group "gf" created with a size hint
 *     gfdb_io_save_trace     :314-427   h5_opencreategroup "/gf/<ixc>" then "<iz>" (:628-647), 1-D dataset "<ig>", attributes
 *                                        "pofs" then "ofs", data, then ONE element of "index" through h5sselect_elements with the
 *                                        Fortran coordinate (ig, iz, ixc) = C coordinate (ixc-1, iz-1, ig-1)
 * The Fortran API hands H5T_NATIVE_REAL / H5T_NATIVE_INTEGER as FILE types, i.e. whatever the writing machine's are; -DBIG
 * stores them as a big-endian machine would (H5T_IEEE_F32BE / H5T_STD_I32BE), -DWIDE as a build with 8-byte default
 * integers would (H5T_STD_I64LE).  The reader has to take all of them.
 *
 * Input (stdin, text): base dt dx dz firstx firstz nchunks nx nxc nz ng ntraces, then per trace:
 *     ix iz ig nstrips npacked  pofs[nstrips]  ofs[nstrips]  packed[npacked]         (ix, iz, ig 1-based; ix global) */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>

#if defined(BIG)
#define FILE_REAL H5T_IEEE_F32BE
#define FILE_INT  H5T_STD_I32BE
#elif defined(WIDE)
#define FILE_REAL H5T_IEEE_F32LE
#define FILE_INT  H5T_STD_I64LE
#else
#define FILE_REAL H5T_NATIVE_FLOAT
#define FILE_INT  H5T_NATIVE_INT
#endif

static void save_scalar(hid_t file, const char *name, hid_t ftype, hid_t mtype, const void *v)      /* :683-741 */
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t d = H5Dcreate2(file, name, ftype, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
    H5Dclose(d);
    H5Sclose(sp);
}

static hid_t opencreategroup(hid_t loc, const char *name, size_t hint)                               /* :628-647 */
{
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t g = H5Gopen2(loc, name, H5P_DEFAULT);
    if (g < 0) g = H5Gcreate1(loc, name, hint);
    return g;
}

int main(void)
{
    char base[2048], fn[4096];
    float dt, dx, dz, firstx, firstz;
    int nchunks, nx, nxc, nz, ng, ntraces;
    if (scanf("%2047s %f %f %f %f %f %d %d %d %d %d %d", base, &dt, &dx, &dz, &firstx, &firstz, &nchunks, &nx, &nxc, &nz, &ng, &ntraces) != 12) return 2;
    /* gfdb_io_create_index */
    snprintf(fn, sizeof fn, "%s.index", base);
    hid_t f = H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return 3;
    save_scalar(f, "dt", FILE_REAL, H5T_NATIVE_FLOAT, &dt);
    save_scalar(f, "dx", FILE_REAL, H5T_NATIVE_FLOAT, &dx);
    save_scalar(f, "dz", FILE_REAL, H5T_NATIVE_FLOAT, &dz);
    save_scalar(f, "firstx", FILE_REAL, H5T_NATIVE_FLOAT, &firstx);
    save_scalar(f, "firstz", FILE_REAL, H5T_NATIVE_FLOAT, &firstz);
    save_scalar(f, "nchunks", FILE_INT, H5T_NATIVE_INT, &nchunks);
    save_scalar(f, "nx", FILE_INT, H5T_NATIVE_INT, &nx);
    save_scalar(f, "nxc", FILE_INT, H5T_NATIVE_INT, &nxc);
    save_scalar(f, "nz", FILE_INT, H5T_NATIVE_INT, &nz);
    save_scalar(f, "ng", FILE_INT, H5T_NATIVE_INT, &ng);
    H5Fclose(f);
    /* gfdb_io_create_chunk: every chunk holds nxc distances, the last one the rest (gfdb.f90:250-257) */
    for (int ic = 1; ic <= nchunks; ic++) {
        snprintf(fn, sizeof fn, "%s.%d.chunk", base, ic);
        f = H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        if (f < 0) return 3;
        const int nxcthis = ic == nchunks ? nx - (ic - 1) * nxc : nxc;
        hsize_t dims[3] = { (hsize_t)nxcthis, (hsize_t)nz, (hsize_t)ng };
        hid_t sp = H5Screate_simple(3, dims, NULL);
        hid_t di = H5Dcreate2(f, "index", H5T_STD_REF_OBJ, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hobj_ref_t *zero = calloc((size_t)nxcthis * nz * ng, sizeof *zero);
        H5Dwrite(di, H5T_STD_REF_OBJ, H5S_ALL, H5S_ALL, H5P_DEFAULT, zero);
        free(zero);
        hid_t g = H5Gcreate1(f, "gf", 64);
        H5Gclose(g);
        H5Dclose(di);
        H5Sclose(sp);
        H5Fclose(f);
    }
    /* gfdb_io_save_trace, one call per stored trace; the file is re-opened per trace as gfdb_save_trace does through its chunk
       cache (gfdb.f90:742-779) */
    for (int t = 0; t < ntraces; t++) {
        int ix, iz, ig, nstrips, npacked;
        if (scanf("%d %d %d %d %d", &ix, &iz, &ig, &nstrips, &npacked) != 5) return 2;
        int *pofs = malloc(sizeof(int) * (size_t)nstrips), *ofs = malloc(sizeof(int) * (size_t)nstrips);
        float *packed = malloc(sizeof(float) * (size_t)npacked);
        for (int i = 0; i < nstrips; i++) if (scanf("%d", &pofs[i]) != 1) return 2;
        for (int i = 0; i < nstrips; i++) if (scanf("%d", &ofs[i]) != 1) return 2;
        for (int i = 0; i < npacked; i++) if (scanf("%f", &packed[i]) != 1) return 2;
        int ic = (ix - 1) / nxc + 1;                                                  /* gfdb.f90:1409-1411 */
        if (ic > nchunks) ic = nchunks;
        const int ixc = ix - (ic - 1) * nxc;
        snprintf(fn, sizeof fn, "%s.%d.chunk", base, ic);
        f = H5Fopen(fn, H5F_ACC_RDWR, H5P_DEFAULT);
        hid_t di = H5Dopen2(f, "index", H5P_DEFAULT);
        char name[64];
        snprintf(name, sizeof name, "/gf/%d", ixc);
        hid_t gd = opencreategroup(f, name, 64);
        snprintf(name, sizeof name, "%d", iz);
        hid_t gz = opencreategroup(gd, name, 64);
        hsize_t n1 = (hsize_t)npacked;
        hid_t dsp = H5Screate_simple(1, &n1, NULL);
        snprintf(name, sizeof name, "%d", ig);
        hid_t d = H5Dcreate2(gz, name, FILE_REAL, dsp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hsize_t ns = (hsize_t)nstrips;
        hid_t asp = H5Screate_simple(1, &ns, NULL);
        hid_t a = H5Acreate2(d, "pofs", FILE_INT, asp, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, H5T_NATIVE_INT, pofs);
        H5Aclose(a);
        a = H5Acreate2(d, "ofs", FILE_INT, asp, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, H5T_NATIVE_INT, ofs);
        H5Aclose(a);
        H5Sclose(asp);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, packed);
        hobj_ref_t ref;
        H5Rcreate(&ref, gz, name, H5R_OBJECT, -1);
        hid_t fsp = H5Dget_space(di);
        hsize_t coord[3] = { (hsize_t)(ixc - 1), (hsize_t)(iz - 1), (hsize_t)(ig - 1) };
        H5Sselect_elements(fsp, H5S_SELECT_SET, 1, coord);
        hsize_t one = 1;
        hid_t msp = H5Screate_simple(1, &one, NULL);
        H5Dwrite(di, H5T_STD_REF_OBJ, msp, fsp, H5P_DEFAULT, &ref);
        H5Sclose(msp); H5Sclose(fsp);
        H5Dclose(d); H5Sclose(dsp); H5Gclose(gd); H5Gclose(gz); H5Dclose(di); H5Fclose(f);
        free(pofs); free(ofs); free(packed);
    }
    return 0;
}
