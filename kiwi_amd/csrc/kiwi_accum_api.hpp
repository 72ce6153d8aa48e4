// kiwi_accum_api.hpp -- what the host library sees of the accumulate kernels: launchers, one set per arithmetic contract.
// The kernels themselves (kiwi_accum.inc) are compiled in translation units of their own -- kiwi_accum.hip, once per
// (kernel family, contract) -- so that the device code of the two contracts is built with different -ffp-contract settings
// and the families compile in parallel.
#pragma once
#include "kiwi_common.hpp"

namespace kiwi {

struct AccumArgs {
    hipStream_t stream;
    int ng;                              // 8 or 10 Green's function components
    bool fuse;                           // comparator inside the kernel's epilogue (FuseParams), no synthetics written
    const float *G; const int2 *span; int pitch;
    const GeoRec *recs; const int *cent_ofs; int isrc0, nrec;
    const RecvDev *recv; float *syn; size_t syn_stride;
    const int *tab; const float *coefs; FuseParams fp;
    const int *pairflag, *synrow, *fam_ofs, *fam_list;
    int compact;                         // `tab` holds geometry_kernel's compact descriptors (four ints per record), not 128-int rows
};

#define KIWI_ACCUM_LAUNCHERS                                                                                                      \
    /* accumulate_kernel: grid (tiles of 1024, receivers, sources) */                                                             \
    void launch_direct(const AccumArgs &a, dim3 grid);                                                                            \
    /* accumulate_grouped_kernel<NG, T, FUSE, RUNS>: T threads = tiles of 4 T samples; runs: run_first[] or null */              \
    void launch_grouped(const AccumArgs &a, dim3 grid, int T, int ntiles, const int *runs, int pairsel, const int *mate,          \
                        const int *mate4);                                                                                        \
    /* accumulate_multi_kernel<NG, FUSE, NS>: NS = 2 / 4 sources per workgroup */                                                 \
    void launch_multi(const AccumArgs &a, dim3 grid, int NS, int ntiles, const int *mate, const int *mate_wider);                \
    /* accumulate_cell_kernel<NG, 256, 2, 0, FUSE> (tile shared by the workgroup) / accumulate_cellw_kernel (tile per wave) */    \
    void launch_cell(const AccumArgs &a, dim3 grid, int ntiles);                                                                  \
    void launch_cellw(const AccumArgs &a, dim3 grid, int ntiles);                                                                 \
    /* largest shift range of a cell group of accumulate_cellw_kernel (cellgroup_kernel cuts there) */                           \
    int cellw_range();

namespace exact { KIWI_ACCUM_LAUNCHERS }
namespace fused { KIWI_ACCUM_LAUNCHERS }
#undef KIWI_ACCUM_LAUNCHERS

} // namespace kiwi
