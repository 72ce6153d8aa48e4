// kiwi_accum.hip -- one translation unit per (kernel family, arithmetic contract) of the accumulate kernels:
//   -DKIWI_FAMILY=1 direct, 2 grouped, 3 multi, 4 cell      -DKIWI_ARITH=0 exact (-ffp-contract=off), 1 fused (-ffp-contract=fast)
// (see the Makefile).  Only the launchers leave this file.
#include "kiwi_accum_api.hpp"

#if KIWI_ARITH == 0
#define KIWI_ARITH_NS exact
#elif KIWI_ARITH == 1
#define KIWI_ARITH_NS fused
#else
#error "KIWI_ARITH: 0 (exact) or 1 (fused)"
#endif

namespace kiwi {
namespace KIWI_ARITH_NS {

#include "kiwi_accum.inc"

#define KIWI_COMMON_ARGS a.G, a.span, a.pitch, a.recs, a.cent_ofs, a.isrc0, a.nrec, a.recv, a.syn, a.syn_stride

#if KIWI_FAMILY == 1
void launch_direct(const AccumArgs &a, dim3 grid)
{
    if (a.ng == 10) hipLaunchKernelGGL(accumulate_kernel<10>, grid, dim3(256), 0, a.stream, KIWI_COMMON_ARGS, a.synrow);
    else            hipLaunchKernelGGL(accumulate_kernel<8>, grid, dim3(256), 0, a.stream, KIWI_COMMON_ARGS, a.synrow);
}
#endif

#if KIWI_FAMILY == 2
template <int NG, int T, bool COMPACT>
static void grouped_tc(const AccumArgs &a, dim3 grid, int ntiles, const int *runs, int pairsel, const int *mate, const int *mate4)
{
#define KIWI_G(FV, RV) hipLaunchKernelGGL((accumulate_grouped_kernel<NG, T, FV, RV, COMPACT>), grid, dim3(T), 0, a.stream, KIWI_COMMON_ARGS, ntiles, \
                                          a.tab, a.coefs, runs, a.fp, a.pairflag, pairsel, mate, mate4, a.synrow, a.fam_ofs, a.fam_list)
    if (a.fuse) { if (runs) KIWI_G(true, true); else KIWI_G(true, false); }
    else        { if (runs) KIWI_G(false, true); else KIWI_G(false, false); }
#undef KIWI_G
}
template <int NG, int T>
static void grouped_t(const AccumArgs &a, dim3 grid, int ntiles, const int *runs, int pairsel, const int *mate, const int *mate4)
{
    if (a.compact) grouped_tc<NG, T, true>(a, grid, ntiles, runs, pairsel, mate, mate4);
    else           grouped_tc<NG, T, false>(a, grid, ntiles, runs, pairsel, mate, mate4);
}
void launch_grouped(const AccumArgs &a, dim3 grid, int T, int ntiles, const int *runs, int pairsel, const int *mate, const int *mate4)
{
    if (a.ng == 10) {
        if (T == 64) grouped_t<10, 64>(a, grid, ntiles, runs, pairsel, mate, mate4);
        else if (T == 128) grouped_t<10, 128>(a, grid, ntiles, runs, pairsel, mate, mate4);
        else grouped_t<10, 256>(a, grid, ntiles, runs, pairsel, mate, mate4);
    } else {
        if (T == 64) grouped_t<8, 64>(a, grid, ntiles, runs, pairsel, mate, mate4);
        else if (T == 128) grouped_t<8, 128>(a, grid, ntiles, runs, pairsel, mate, mate4);
        else grouped_t<8, 256>(a, grid, ntiles, runs, pairsel, mate, mate4);
    }
}
#endif

#if KIWI_FAMILY == 3
template <int NG, int NS, bool COMPACT>
static void multi_t(const AccumArgs &a, dim3 grid, int ntiles, const int *mate, const int *wider)
{
    if (a.fuse) hipLaunchKernelGGL((accumulate_multi_kernel<NG, true, NS, COMPACT>), grid, dim3(256), 0, a.stream, KIWI_COMMON_ARGS, ntiles, a.tab, a.coefs, a.fp,
                                   a.pairflag, mate, wider);
    else        hipLaunchKernelGGL((accumulate_multi_kernel<NG, false, NS, COMPACT>), grid, dim3(256), 0, a.stream, KIWI_COMMON_ARGS, ntiles, a.tab, a.coefs, a.fp,
                                   a.pairflag, mate, wider);
}
template <int NG, int NS>
static void multi_c(const AccumArgs &a, dim3 grid, int ntiles, const int *mate, const int *wider)
{
    if (a.compact) multi_t<NG, NS, true>(a, grid, ntiles, mate, wider); else multi_t<NG, NS, false>(a, grid, ntiles, mate, wider);
}
void launch_multi(const AccumArgs &a, dim3 grid, int NS, int ntiles, const int *mate, const int *wider)
{
    if (a.ng == 10) { if (NS == 4) multi_c<10, 4>(a, grid, ntiles, mate, wider); else multi_c<10, 2>(a, grid, ntiles, mate, wider); }
    else            { if (NS == 4) multi_c<8, 4>(a, grid, ntiles, mate, wider); else multi_c<8, 2>(a, grid, ntiles, mate, wider); }
}
#endif

#if KIWI_FAMILY == 4
template <int NG, bool COMPACT>
static void cell_t(const AccumArgs &a, dim3 grid, int ntiles, bool per_wave)
{
#define KIWI_C(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, a.stream, KIWI_COMMON_ARGS, ntiles, a.tab, a.coefs, a.fp, a.pairflag, a.synrow, \
                                          a.fam_ofs, a.fam_list)
    if (per_wave) { if (a.fuse) KIWI_C((accumulate_cellw_kernel<NG, true, COMPACT>)); else KIWI_C((accumulate_cellw_kernel<NG, false, COMPACT>)); }
    else          { if (a.fuse) KIWI_C((accumulate_cell_kernel<NG, 256, 2, 0, true, COMPACT>)); else KIWI_C((accumulate_cell_kernel<NG, 256, 2, 0, false, COMPACT>)); }
#undef KIWI_C
}
template <int NG>
static void cell_c(const AccumArgs &a, dim3 grid, int ntiles, bool per_wave)
{
    if (a.compact) cell_t<NG, true>(a, grid, ntiles, per_wave); else cell_t<NG, false>(a, grid, ntiles, per_wave);
}
void launch_cell(const AccumArgs &a, dim3 grid, int ntiles) { if (a.ng == 10) cell_c<10>(a, grid, ntiles, false); else cell_c<8>(a, grid, ntiles, false); }
void launch_cellw(const AccumArgs &a, dim3 grid, int ntiles) { if (a.ng == 10) cell_c<10>(a, grid, ntiles, true); else cell_c<8>(a, grid, ntiles, true); }
int cellw_range() { return kCellwRange; }
#endif

} // namespace KIWI_ARITH_NS
} // namespace kiwi
