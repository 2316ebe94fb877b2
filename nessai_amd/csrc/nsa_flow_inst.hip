// One instantiation of the fused flow kernel, selected at compile time:
//   -DNSA_HT=<hidden tiles> -DNSA_TKIND=<NSA_T_*> -DNSA_BINS=<K or 0>
// The Makefile builds the supported matrix as separate objects (parallel
// compile); nsa_api.hip dispatches on the flow image header.
#include "nsa_flow_kernel.h"

#if !defined(NSA_HT) || !defined(NSA_TKIND) || !defined(NSA_BINS)
#error "NSA_HT, NSA_TKIND and NSA_BINS must be defined"
#endif
// weights-resident shapes (waves per SIMD, row tiles per wave) beside the always-built (2, 1); the same list
// as nsa_resident_candidates in nsa_api.hip.  64-wide conditioners do not fit 128 registers (4 waves per SIMD:
// ~70 spilled) and two tiles per wave at 3 waves per SIMD measured slowest everywhere (and spill): not built.
#ifndef NSA_RES_GEOMETRIES
#if NSA_HT == 1
#define NSA_RES_GEOMETRIES NSA_RES_GEO(3, 1) NSA_RES_GEO(4, 1) NSA_RES_GEO(2, 2)
#else
#define NSA_RES_GEOMETRIES NSA_RES_GEO(3, 1) NSA_RES_GEO(2, 2)
#endif
#endif

#if NSA_TKIND == 1 && NSA_BINS == 99  // the generic instance: any num_bins <= 16 on padded 16-bin parameter sets
using TP = NsaRQ<16, true>;
#elif NSA_TKIND == 1
using TP = NsaRQ<NSA_BINS>;
#elif NSA_TKIND == 2
using TP = NsaAffine;
#elif NSA_TKIND == 3
using TP = NsaAdditive;
#else
#error "unknown NSA_TKIND"
#endif

#define NSA_CAT_(a, b, c, d, e, f) a##b##c##d##e##f
#define NSA_CAT(a, b, c, d, e, f) NSA_CAT_(a, b, c, d, e, f)
#define NSA_LAUNCH_NAME NSA_CAT(nsa_launch_h, NSA_HT, _t, NSA_TKIND, _k, NSA_BINS)

// kernel of one (direction, precision, waves per SIMD, row tiles per wave)
template <int PREC, int OCC, int RT>
static auto nsa_pick(bool inv) {
  return inv ? nsa_flow_kernel<NSA_HT, TP, true, PREC, OCC, RT> : nsa_flow_kernel<NSA_HT, TP, false, PREC, OCC, RT>;
}

// weights-resident instances (<= 64-wide conditioners): no barrier in the tile loop, so the geometries go up
// to 4 waves per SIMD; NSA_RES_GEOMETRIES (Makefile) lists the (waves per SIMD, row tiles) pairs that are built
#if NSA_HT <= 2
template <int PREC>
static auto nsa_pick_res(const NsaLaunch* p, bool inv) {
  auto kern = inv ? nsa_flow_kernel<NSA_HT, TP, true, PREC, 2, 1, true> : nsa_flow_kernel<NSA_HT, TP, false, PREC, 2, 1, true>;
  if (p->occupancy != 2 || p->row_tiles != 1) kern = nullptr;  // a shape that is not built must not run another one's code
  // (exact f32: one row tile per wave only, as in the staged geometries)
#define NSA_RES_GEO(occ, rt)                                                      \
  if constexpr (rt == 1 || PREC != NSA_PREC_F32)                                  \
    if (p->occupancy == occ && p->row_tiles == rt)                                \
      kern = inv ? nsa_flow_kernel<NSA_HT, TP, true, PREC, occ, rt, true> : nsa_flow_kernel<NSA_HT, TP, false, PREC, occ, rt, true>;
  NSA_RES_GEOMETRIES
#undef NSA_RES_GEO
  return kern;
}
#endif

// the split 16-bit modes share their geometries
template <int PREC>
static auto nsa_pick16(const NsaLaunch* p, bool inv) {
  auto kern = nsa_pick<PREC, 2, 1>(inv);
#if NSA_HT <= 2
  // three waves per SIMD: the hidden state of a <= 64-wide conditioner fits 168 VGPRs
  if (p->occupancy == 3) kern = nsa_pick<PREC, 3, 1>(inv);
  // two row tiles per wave, 8-wave workgroups at two waves per SIMD (not in exact f32: the matrix pipe is
  // the limit there and the variant measured 4.41 ms against 4.36)
  if (p->row_tiles == 2) kern = nsa_pick<PREC, 2, 2>(inv);
#endif
  return kern;
}

// conditional flows (context columns + gated residual blocks): the staged one-tile geometry, every precision
template <int PREC>
static auto nsa_pick_ctx(bool inv) {
  return inv ? nsa_flow_kernel<NSA_HT, TP, true, PREC, 2, 1, false, true> : nsa_flow_kernel<NSA_HT, TP, false, PREC, 2, 1, false, true>;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t NSA_LAUNCH_NAME(
    const NsaLaunch* p, unsigned grid, unsigned grid_y, unsigned block, size_t lds, hipStream_t stream) {
  const bool inv = p->direction == NSA_DIR_INVERSE;
  auto kern = nsa_pick<NSA_PREC_F32, 2, 1>(inv);
  if (p->ext) {
    if (p->resident || p->occupancy != 2 || p->row_tiles != 1) return hipErrorInvalidConfiguration;
    kern = nsa_pick_ctx<NSA_PREC_F32>(inv);
    if (p->precision == NSA_PREC_BF16X3) kern = nsa_pick_ctx<NSA_PREC_BF16X3>(inv);
    else if (p->precision == NSA_PREC_F16X3) kern = nsa_pick_ctx<NSA_PREC_F16X3>(inv);
  } else
#if NSA_HT <= 2
  if (p->resident) {
    kern = nsa_pick_res<NSA_PREC_F32>(p, inv);
    if (p->precision == NSA_PREC_BF16X3) kern = nsa_pick_res<NSA_PREC_BF16X3>(p, inv);
    else if (p->precision == NSA_PREC_F16X3) kern = nsa_pick_res<NSA_PREC_F16X3>(p, inv);
  } else
#endif
  if (p->precision == NSA_PREC_BF16X3) kern = nsa_pick16<NSA_PREC_BF16X3>(p, inv);
  else if (p->precision == NSA_PREC_F16X3) kern = nsa_pick16<NSA_PREC_F16X3>(p, inv);
#if NSA_HT <= 2
  else if (p->occupancy == 3) kern = nsa_pick<NSA_PREC_F32, 3, 1>(inv);
#endif
  if (kern == nullptr) return hipErrorInvalidConfiguration;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid, grid_y), dim3(block), lds, stream, *p);
  return hipGetLastError();
}
