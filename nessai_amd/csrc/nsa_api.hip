// C-ABI entry points (include/nessai_amd.h) and kernel dispatch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/nessai_amd.h"
#include "nsa_flow_kernel.h"

// (each with its own guard: a variant build may pass any one of them with -D, scripts/build_variant.sh)
#ifndef NSA_DEFAULT_OCC
#define NSA_DEFAULT_OCC 2
#endif
#ifndef NSA_DEFAULT_LOADERS
#define NSA_DEFAULT_LOADERS 3  // loader waves per workgroup of the small-batch geometry (NsaLaunch::loader): 1 computing wave + 3 = one wave per SIMD
#endif
#ifndef NSA_LOADER_MAX_TILES
#define NSA_LOADER_MAX_TILES 5  // ... up to this many tiles per CU (as many tiles per workgroup, sharing the loaders)
#endif
#ifndef NSA_TEAM_MIN_HT
#define NSA_TEAM_MIN_HT 2  // conditioners of >= 64 features take the loader geometry on small batches (nsa_launch_flow)
#endif
#ifndef NSA_DEFAULT_SKEW
#define NSA_DEFAULT_SKEW 1
#endif

// ---------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------

static thread_local char g_err[512] = "";

static int nsa_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* nsa_last_error(void) { return g_err; }

static thread_local int g_last_geometry[6] = {0, 0, 0, 0, 0, 0};
extern "C" void nsa_last_geometry(int* waves_per_simd, int* waves_per_group, int* groups) {
  if (waves_per_simd != nullptr) *waves_per_simd = g_last_geometry[0];
  if (waves_per_group != nullptr) *waves_per_group = g_last_geometry[1];
  if (groups != nullptr) *groups = g_last_geometry[2];
}
extern "C" int nsa_last_row_tiles(void) { return g_last_geometry[3]; }
extern "C" int nsa_last_resident(void) { return g_last_geometry[4]; }
static thread_local int g_last_ring = 0;
extern "C" int nsa_last_team(void) { return g_last_geometry[5]; }
extern "C" int nsa_last_ring(void) { return g_last_geometry[5] ? g_last_ring : 0; }
extern "C" int nsa_version(void) { return NSA_VERSION; }

// ---------------------------------------------------------------------------
// kernel instance table (objects built from nsa_flow_inst.hip by the Makefile;
// NSA_INSTANCES is generated there as X(ht, kind, bins) entries)
// ---------------------------------------------------------------------------

typedef hipError_t (*nsa_launch_fn)(const NsaLaunch*, unsigned, unsigned, unsigned, size_t, hipStream_t);

#define X(ht, kind, bins) \
  extern "C" hipError_t nsa_launch_h##ht##_t##kind##_k##bins(const NsaLaunch*, unsigned, unsigned, unsigned, size_t, hipStream_t);
NSA_INSTANCES
#undef X

struct NsaInstance {
  int ht, kind, bins;
  nsa_launch_fn fn;
};

static const NsaInstance g_instances[] = {
#define X(ht, kind, bins) {ht, kind, bins, nsa_launch_h##ht##_t##kind##_k##bins},
    NSA_INSTANCES
#undef X
};

#define NSA_GENERIC_BINS 99  // instance key of NsaRQ<16, true>: any num_bins from 2 to 16 (nsa_device.h)

static nsa_launch_fn nsa_find(int ht, int kind, int bins) {
  if (kind != NSA_T_RQ) bins = 0;
  for (size_t i = 0; i < sizeof(g_instances) / sizeof(g_instances[0]); ++i)
    if (g_instances[i].ht == ht && g_instances[i].kind == kind && g_instances[i].bins == bins)
      return g_instances[i].fn;
  // no exact spline instance: the generic one runs the flow on padded parameter sets (the packer lays them out)
  if (kind == NSA_T_RQ && bins >= 2 && bins <= 16 && bins != NSA_GENERIC_BINS) return nsa_find(ht, kind, NSA_GENERIC_BINS);
  return nullptr;
}

// 1 when a spline flow of `bins` bins runs on the generic (padded) instance rather than an exact one
extern "C" int nsa_flow_generic_bins(int hidden_tiles, int bins) {
  for (size_t i = 0; i < sizeof(g_instances) / sizeof(g_instances[0]); ++i)
    if (g_instances[i].ht == hidden_tiles && g_instances[i].kind == NSA_T_RQ && g_instances[i].bins == bins) return 0;
  return nsa_find(hidden_tiles, NSA_T_RQ, bins) != nullptr ? 1 : 0;
}

extern "C" int nsa_flow_supported(int hidden_tiles, int transform, int bins) {
  return nsa_find(hidden_tiles, transform, bins) != nullptr ? 1 : 0;
}

// ---------------------------------------------------------------------------
// launch geometry
// ---------------------------------------------------------------------------

// per-device geometry, filled on first use of each device (a process may drive several GPUs)
struct NsaDeviceInfo {
  int num_cus;
  size_t max_lds;
};
#define NSA_MAX_DEVICES 64
static NsaDeviceInfo g_devices[NSA_MAX_DEVICES];
static std::mutex g_devices_mutex;

static int nsa_device_info(NsaDeviceInfo& out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= NSA_MAX_DEVICES) return nsa_fail(NSA_E_HIP, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_devices_mutex);
  if (g_devices[dev].num_cus == 0) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    g_devices[dev].num_cus = prop.multiProcessorCount;
    g_devices[dev].max_lds = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
  }
  out = g_devices[dev];
  return NSA_OK;
}

// Launch geometry of the flow kernel, as a code: waves per SIMD | (row tiles per wave - 1) << 5.
//   2     workgroups of 4 waves, two per CU (the default)
//   3     ONE workgroup of 12 waves per CU at 3 waves per SIMD (<= 168 VGPRs): a third of the stage
//         copies per row and one more wave per SIMD to cover the stage barriers
//   0x22  ONE workgroup of 8 waves per CU at 2 waves per SIMD, each wave taking TWO 32-row tiles through
//         every layer together: a weight operand read from LDS feeds 6 MFMAs instead of 3, a stage is
//         copied once per 512 rows, and every wave has two independent accumulator / spline chains
//         (config 3: 1.70 ms per 1e6 rows where the other two take 1.93 on the same box)
// The last two are for big split-bf16 batches of a narrow conditioner (HT <= 2): nsa_big_batch_geometry.
// NESSAI_AMD_OCC=2|3|2x2 overrides (read per call: the benchmark scripts flip it).
#define NSA_GEO_2X2 0x22

// Big batches: the geometry with the least estimated time = whole rounds of the grid x tiles per round
// x relative time per tile (steady-state draw kernel of config 3 on three MI355X boxes: 1.67-1.70 ms
// per 1e6 rows with two tiles per wave, 1.75-1.93 with 12 waves, 1.75-1.94 with 2 x 4 waves).  The
// rounds matter at the lower end: 6144 tiles are 1.5 rounds of 256 x 16 tiles but exactly 2 of 256 x 12.
static int nsa_big_batch_geometry(long long tiles, int num_cus) {
  const int geo[3] = {NSA_GEO_2X2, 3, 2};
  const long long per_round[3] = {16LL * num_cus, 12LL * num_cus, 8LL * num_cus};
  const double per_tile[3] = {1.00, 1.06, 1.09};
  int best = NSA_GEO_2X2;
  double best_t = 0.0;
  for (int c = 0; c < 3; ++c) {
    const double t = (double)((tiles + per_round[c] - 1) / per_round[c]) * (double)per_round[c] * per_tile[c];
    if (c == 0 || t < best_t) {
      best = geo[c];
      best_t = t;
    }
  }
  return best;
}
static int nsa_occupancy(int ht, int precision, long long n_tiles, int n_flows, int num_cus,
                         int waves_per_group) {
  const char* e = getenv("NESSAI_AMD_OCC");
  if (e != nullptr && e[0] == '2' && e[1] == 'x' && e[2] == '2')
    return (ht <= 2 && precision != NSA_PREC_F32 && waves_per_group <= 0) ? NSA_GEO_2X2 : NSA_DEFAULT_OCC;
  if (e != nullptr && (e[0] == '2' || e[0] == '3')) return (e[0] == '3' && ht <= 2) ? 3 : 2;
  if (waves_per_group > 0) return (waves_per_group > 8 && ht <= 2) ? 3 : NSA_DEFAULT_OCC;
  // (3 = "big batch": nsa_launch_flow replaces it by nsa_big_batch_geometry's estimate; from one full round
  // of the default geometry upwards - 2048 tiles on 256 CUs - the estimate is the measured order:
  // scripts/bench_geometry_sizes.py, 62 500 to 1e6 rows)
  if (ht <= 2 && precision != NSA_PREC_F32 && n_tiles * n_flows >= 8LL * num_cus) return 3;
  return NSA_DEFAULT_OCC;
}

static int nsa_check_header(const uint32_t* h) {
  if (h == nullptr) return nsa_fail(NSA_E_ARG, "header is NULL");
  if (h[NSA_H_MAGIC] != NSA_MAGIC) return nsa_fail(NSA_E_IMAGE, "bad flow image magic 0x%08x", h[NSA_H_MAGIC]);
  if (h[NSA_H_VERSION] != NSA_VERSION)
    return nsa_fail(NSA_E_IMAGE, "flow image version %u, library expects %d", h[NSA_H_VERSION], NSA_VERSION);
  if (h[NSA_H_D] < 2 || h[NSA_H_D] > 1024) return nsa_fail(NSA_E_IMAGE, "unsupported feature count %u", h[NSA_H_D]);
  if ((h[NSA_H_MAX_STAGE] & 255u) != 0) return nsa_fail(NSA_E_IMAGE, "stage size not a multiple of 256 words");
  return NSA_OK;
}

static int nsa_launch_flow(const uint32_t* h, NsaLaunch& p, int waves_per_group, int max_groups,
                           hipStream_t stream, int n_flows = 1, int force_occ = 0);

// NESSAI_AMD_TUNE=1 (or "verbose") replaces that static rule by a measurement: the first big launch of a
// (device, kernel instance, mode, program) times the three candidates on the caller's stream - the same
// launch, whose output does not depend on the geometry, with a private status buffer and without the
// device-side loop control; every candidate twice untimed, then five rounds, best time of each - and the
// choice is kept for the process.  Off by default: a probe at the start of a process sees the candidates
// 0.05-0.15 ms slower than they run in steady state and closer together than they are (1.81 / 1.82 / 1.80 ms
// for kernels that then run at 1.67 / 1.76 / 1.75), so it picked the slower geometry on two boxes of three.
// A stream that is being captured keeps the static rule.
struct NsaTuneEntry {
  int device, ht, kind, bins, mode, n_stages, occ;
};
static std::mutex g_tune_mutex;
static std::vector<NsaTuneEntry> g_tune;

static int nsa_tuned_occupancy(const uint32_t* h, const NsaLaunch& p, int max_groups, hipStream_t stream,
                               int n_flows, int fallback) {
  const char* env = getenv("NESSAI_AMD_TUNE");
  if (env == nullptr || (env[0] != '1' && env[0] != 'v')) return fallback;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return fallback;
  const bool inv = p.direction == NSA_DIR_INVERSE;
  const NsaTuneEntry key = {device, (int)h[NSA_H_HT], (int)h[NSA_H_TRANSFORM], (int)h[NSA_H_BINS], p.mode,
                            (int)h[inv ? NSA_H_NSTAGES_INV : NSA_H_NSTAGES_FWD], 0};
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  for (const NsaTuneEntry& e : g_tune)
    if (e.device == key.device && e.ht == key.ht && e.kind == key.kind && e.bins == key.bins && e.mode == key.mode &&
        e.n_stages == key.n_stages)
      return e.occ;
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone) return fallback;
  int* status = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (hipMalloc((void**)&status, 4 * sizeof(int)) != hipSuccess) return fallback;
  int best = fallback;
  if (hipMemsetAsync(status, 0, 4 * sizeof(int), stream) == hipSuccess && hipEventCreate(&t0) == hipSuccess &&
      hipEventCreate(&t1) == hipSuccess) {
    NsaLaunch q = p;
    q.status = status;
    q.stop = nullptr;
    float best_ms = 0.0f;
    bool ok = true;
    const int candidates[3] = {NSA_GEO_2X2, 3, 2};
    float best_of[3] = {0.0f, 0.0f, 0.0f};
    for (int warm = 0; warm < 2 && ok; ++warm)
      for (int c = 0; c < 3 && ok; ++c) ok = nsa_launch_flow(h, q, 0, max_groups, stream, n_flows, candidates[c]) == NSA_OK;
    for (int round = 0; round < 5 && ok; ++round)
      for (int c = 0; c < 3 && ok; ++c) {
        ok = hipEventRecord(t0, stream) == hipSuccess;
        ok = ok && nsa_launch_flow(h, q, 0, max_groups, stream, n_flows, candidates[c]) == NSA_OK;
        ok = ok && hipEventRecord(t1, stream) == hipSuccess && hipEventSynchronize(t1) == hipSuccess;
        float ms = 0.0f;
        ok = ok && hipEventElapsedTime(&ms, t0, t1) == hipSuccess;
        if (ok && (round == 0 || ms < best_of[c])) best_of[c] = ms;
      }
    for (int c = 0; c < 3 && ok; ++c)
      if (c == 0 || best_of[c] < best_ms) {
        best = candidates[c];
        best_ms = best_of[c];
      }
    if (ok && env != nullptr && env[0] == 'v')  // NESSAI_AMD_TUNE=verbose
      fprintf(stderr, "nessai_amd: geometry timing (ms) 2x2 %.3f, 12 waves %.3f, 2 x 4 waves %.3f -> %s\n",
              best_of[0], best_of[1], best_of[2], best == NSA_GEO_2X2 ? "2x2" : (best == 3 ? "3" : "2"));
    if (ok) {
      NsaTuneEntry e = key;
      e.occ = best;
      g_tune.push_back(e);
    } else {
      best = fallback;
    }
  }
  if (t0 != nullptr) (void)hipEventDestroy(t0);
  if (t1 != nullptr) (void)hipEventDestroy(t1);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(status);
  return best;
}

static int nsa_launch_flow(const uint32_t* h, NsaLaunch& p, int waves_per_group, int max_groups,
                           hipStream_t stream, int n_flows, int force_occ) {
  int rc = nsa_check_header(h);
  if (rc != NSA_OK) return rc;
  NsaDeviceInfo dev;
  rc = nsa_device_info(dev);
  if (rc != NSA_OK) return rc;
  const int g_num_cus = dev.num_cus;
  const size_t g_max_lds = dev.max_lds;
  if (p.n < 0) return nsa_fail(NSA_E_ARG, "negative row count");
  if (p.n == 0) return NSA_OK;
  nsa_launch_fn fn = nsa_find((int)h[NSA_H_HT], (int)h[NSA_H_TRANSFORM], (int)h[NSA_H_BINS]);
  if (fn == nullptr)
    return nsa_fail(NSA_E_ARG, "no kernel instance for hidden_tiles=%u transform=%u bins=%u",
                    h[NSA_H_HT], h[NSA_H_TRANSFORM], h[NSA_H_BINS]);
  if (h[NSA_H_PREC] != NSA_PREC_F32 && h[NSA_H_PREC] != NSA_PREC_BF16X3 && h[NSA_H_PREC] != NSA_PREC_F16X3)
    return nsa_fail(NSA_E_IMAGE, "unknown precision code %u", h[NSA_H_PREC]);
  p.precision = (int)h[NSA_H_PREC];
  {
    p.prio = -1;  // decided with the geometry below; NESSAI_AMD_PRIO=0|1|2 pins it (NsaLaunch::prio)
  }
  const int D = (int)h[NSA_H_D];
  const bool wide = h[NSA_H_HT] >= NSA_WIDE_HT;  // 256-wide conditioners: instances built for one wave per SIMD, <= 4 waves per workgroup
  const int C = (int)h[NSA_H_CTX];  // context features of a conditional flow
  if (C < 0 || C > 256) return nsa_fail(NSA_E_IMAGE, "unsupported number of context features %d", C);
  if ((C > 0) != (p.ctx_in != nullptr))
    return nsa_fail(NSA_E_ARG, C > 0 ? "the flow is conditional: a context array is required (nsa_flow_run_ctx)"
                                     : "the flow takes no context");
  if (C > 0 && (n_flows != 1 || p.mode == NSA_MODE_POPULATE))
    return nsa_fail(NSA_E_ARG, "conditional flows run through nsa_flow_run_ctx only");
  // the extended-conditioner instances (context and / or batch norm within the residual blocks): staged, one tile per wave
  p.ext = (C > 0 || h[NSA_H_BNW] != 0) ? 1 : 0;
  const int XS = (D + 2 + C) | 1;  // row stride of the x tiles (nsa_flow_kernel.h)
  const long long n_tiles = (p.n + NSA_ROWS - 1) / NSA_ROWS;
#ifdef NSA_TIMELINE
  {
    const char* tl = getenv("NSA_TL_PTR");  // diagnostic build: device uint64[16] the waves add their section cycles to
    p.dbg = tl != nullptr ? (unsigned long long*)strtoull(tl, nullptr, 0) : nullptr;
  }
#endif
  // Weights-resident geometry (NsaResidentPipe, nsa_flow_kernel.h): the direction's whole program in LDS, ONE
  // workgroup per CU, no barrier in the tile loop.  Taken whenever the program fits beside the x tiles of a
  // useful number of waves; NESSAI_AMD_RESIDENT=0 switches it off (the staged geometries, which give the same
  // rows bit for bit: tests/test_gpu_row_tiles.py), NESSAI_AMD_RESIDENT="<waves per SIMD>x<tiles per wave>"
  // pins one of the built candidates.
  const bool inv = p.direction == NSA_DIR_INVERSE;
  const size_t res_bytes = (((size_t)h[inv ? NSA_H_RES_INV : NSA_H_RES_FWD] + 255) & ~(size_t)255) * 4;  // whole 1 KiB pieces
  // + the populate epilogue's per-column constants: 4 double tables and one int table (nsa_flow_kernel.h)
  const size_t epi_bytes = (p.mode == NSA_MODE_POPULATE) ? (size_t)((D + 3) & ~3) * (4 * 8 + 4) : 0;
  // + the direction's program (stage table + op table), read from LDS at every stage / op
  const size_t prog_bytes = (((size_t)2 * h[inv ? NSA_H_NSTAGES_INV : NSA_H_NSTAGES_FWD] +
                              (size_t)NSA_OP_WORDS * h[inv ? NSA_H_NOPS_INV : NSA_H_NOPS_FWD] + 3) & ~(size_t)3) * 4;
  // (NESSAI_AMD_OCC pins a STAGED geometry)
  if (p.ext) force_occ = NSA_DEFAULT_OCC;  // extended conditioner: the staged one-tile geometry (nsa_flow_inst.hip)
  if (force_occ == 0 && h[NSA_H_HT] <= 2 && res_bytes > 0 && getenv("NESSAI_AMD_OCC") == nullptr) {
    const char* env = getenv("NESSAI_AMD_RESIDENT");
    int pin_occ = 0, pin_rt = 0;
    bool on = true;
    if (env != nullptr && env[0] == '0') on = false;
    else if (env != nullptr && env[0] >= '2' && env[0] <= '4' && env[1] == 'x' && (env[2] == '1' || env[2] == '2')) {
      pin_occ = env[0] - '0';
      pin_rt = env[2] - '0';
    }
    const size_t fixed = res_bytes + epi_bytes + prog_bytes;
    // (x tiles + one parked float per lane and tile, nsa_flow_kernel.h)
    // (+ one double per wave: the lean pass's running weight maximum)
    auto res_lds = [&](int w, int r) { return ((((size_t)w * r * NSA_ROWS * XS + 3) & ~(size_t)3)) * 4 + (size_t)w * r * 256 + (size_t)w * 8 + fixed; };
    // candidates in the order of their measured time per row on big batches (profiles/r04_resident_geometries.jsonl)
    static const int cand[4][2] = {{4, 1}, {3, 1}, {2, 2}, {2, 1}};
    for (int c = 0; c < 4 && on; ++c) {
      int o = cand[c][0], r = cand[c][1];
      if (o == 4 && h[NSA_H_HT] != 1) continue;  // built for 32-wide conditioners only (nsa_flow_inst.hip)
      if (pin_occ != 0 && (o != pin_occ || r != pin_rt)) continue;
      if (r == 2 && p.precision == NSA_PREC_F32) continue;  // exact f32 keeps one tile per wave, as in the staged geometries
      int w = waves_per_group > 0 ? waves_per_group : 4 * o;
      if (w > 4 * o) w = 4 * o;
      if (waves_per_group <= 0 && pin_occ == 0) {
        // batches that do not fill the chip: one tile per wave and just the waves per workgroup that take
        // the batch in ONE round over all CUs (a tile's trip through the flow is tens of microseconds of
        // latency: a second round for a few left-over tiles doubles a small launch)
        if (r == 2 && n_tiles * n_flows < (long long)w * r * g_num_cus) continue;
        if (n_tiles * n_flows < (long long)w * g_num_cus) {
          const long long per_cu = (n_tiles * n_flows + g_num_cus - 1) / g_num_cus;
          w = (int)(per_cu < 1 ? 1 : per_cu);
        }
      }
      const int w_asked = w;
      while (w > 1 && res_lds(w, r) > g_max_lds) --w;  // fewer waves beside a bigger program
      if (res_lds(w, r) > g_max_lds) continue;
      // ... but a program that leaves room for fewer than two tiles per SIMD is better off staged: a 135 KB
      // spline flow at five waves per CU measured 0.69 ms per 1e6 rows resident against 0.53 staged, a 118 KB
      // one at four waves 0.91 against 0.65 (profiles/r04_c3_coupling_resident.jsonl)
      if (pin_occ == 0 && w < w_asked && w * r < 8) continue;
      p.resident = 1;
      {
        const char* pr = getenv("NESSAI_AMD_PRIO");
        p.prio = pr != nullptr ? atoi(pr) : 0;
      }
      p.occupancy = o;
      p.row_tiles = r;
      p.skew = 0;
      const long long n_groups = (n_tiles + (long long)w * r - 1) / ((long long)w * r);
      long long grid = (g_num_cus + n_flows - 1) / n_flows;
      if (max_groups > 0 && grid > max_groups) grid = max_groups;
      if (grid > n_groups) grid = n_groups;
      hipError_t e = fn(&p, (unsigned)grid, (unsigned)n_flows, (unsigned)(w * 64), res_lds(w, r), stream);
      if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "flow kernel launch (resident): %s", hipGetErrorString(e));
      g_last_geometry[0] = o;
      g_last_geometry[3] = r;
      g_last_geometry[1] = w;
      g_last_geometry[2] = (int)grid;
      g_last_geometry[4] = 1;
      g_last_geometry[5] = 0;
      return NSA_OK;
    }
  }
  // Small-batch geometry (NsaLaunch::team / ring / loader, nsa_flow_kernel.h): a batch with fewer tiles than the chip
  // has CUs is pure latency - one workgroup per tile walks the whole program, and requesting the stage copies cost the
  // computing wave a quarter of its time (60 - 200 cycles per 1 KiB piece).  Such a launch gets, per tile, ONE
  // computing wave, THREE loader waves that only request the copies (a wave has at most 63 vector-memory operations in
  // flight - the counter's width - and a copy takes microseconds to land on a chip where every CU streams the same
  // program: one loader's window was the bound, 0.335 ms; two 0.276, three 0.266, four 0.273) and a ring of up to
  // eight stage buffers.  Config 4's 5050 rows (158 tiles, 128-wide conditioner, 4 MB program) 0.40 -> 0.27 ms, a
  // 192-wide flow 0.38 -> 0.27 (one loader), config 3's flow over 2000 rows 0.149 -> 0.136 (one loader; gpurun_out
  // r05n / r05t; 32-wide conditioners gain nothing and keep the staged geometry).  Teams of 2 / 4 / 8 computing waves
  // per tile (NESSAI_AMD_TEAM) share the trunk's work badly - the chunk stages still run one after the other - and
  // measured no faster than one wave: kept as pinned options, same rows bit for bit (tests/test_gpu_team.py).
  // NESSAI_AMD_TEAM=0 switches the geometry off, NESSAI_AMD_LOADER=0 ... 4 sets the loaders, NESSAI_AMD_RING the ring.
  {
    int team = 0;
    const char* te = getenv("NESSAI_AMD_TEAM");
    if (te != nullptr) team = atoi(te);
    else if (h[NSA_H_HT] >= NSA_TEAM_MIN_HT && n_tiles * n_flows <= (long long)NSA_LOADER_MAX_TILES * g_num_cus &&
             waves_per_group <= 0 && force_occ == 0 && getenv("NESSAI_AMD_OCC") == nullptr)
      team = 1;
    if (team == 1 || team == 2 || team == 4 || team == 8) {  // (1: a wave per tile and workgroup, on the deep stage ring)
      const size_t one_stage = (size_t)h[NSA_H_MAX_STAGE] * 4;
      // one team per workgroup: its tile, the parked floats / running maximum / bad-root counts of its waves, the
      // record of Jacobian terms, the populate constants and the program - and as many stage buffers as fit beside
      // them, up to eight (NsaRingPipe: the copies of a small batch are requested ring - 1 stages ahead)
      // (+ the loader: one more wave that only requests the stage copies, NsaLaunch::loader; NESSAI_AMD_LOADER=0: none)
      const char* le = getenv("NESSAI_AMD_LOADER");
      int loader = le != nullptr ? atoi(le) : NSA_DEFAULT_LOADERS;
      loader = loader < 0 ? 0 : (loader > 4 ? 4 : loader);
      const int max_waves = wide ? 4 : 8;
      // tiles (teams) per workgroup: as many as it takes for ONE round of workgroups over the chip - they share the
      // loaders and every stage copy (config 4's flow: 10000 rows 0.42 -> 0.29 ms at two tiles per workgroup, 20000 rows
      // 0.42 -> 0.32 at four, 40400 rows 0.42 -> 0.40 at five; gpurun_out r05v) - NESSAI_AMD_TILES pins the number
      int tpg = (int)((n_tiles * n_flows + g_num_cus - 1) / g_num_cus);
      tpg = tpg < 1 ? 1 : tpg;
      bool fits = true;
      {
        const char* tp = getenv("NESSAI_AMD_TILES");
        const int want = tpg;
        if (tp != nullptr && atoi(tp) >= 1 && atoi(tp) <= 7) tpg = atoi(tp);
        while (tpg > 1 && team * tpg + (loader > 0 ? 1 : 0) > max_waves) --tpg;
        // (the library's own choice only where ONE round of workgroups covers the batch; a pinned team that does not
        // fit the instance's workgroup at all keeps the staged geometry)
        fits = team * tpg <= max_waves && (te != nullptr || tp != nullptr || tpg >= want);
      }
      if (team * tpg + loader > max_waves) loader = max_waves - team * tpg;  // (the instances' 512 / 256 threads)
      if (loader < 0) loader = 0;
      const int waves = team * tpg + loader;
      const size_t fixed = (((size_t)tpg * NSA_ROWS * XS + 3) & ~(size_t)3) * 4 + (size_t)waves * (256 + 8 + 256) +
                           (size_t)tpg * 64 * ((D + 1) / 2 + 16) * 4 + epi_bytes + prog_bytes;
      int ring = one_stage > 0 && g_max_lds > fixed ? (int)((g_max_lds - fixed) / one_stage) : 0;
      if (ring > 8) ring = 8;
      {
        const char* re = getenv("NESSAI_AMD_RING");  // pins the ring's depth (2 = the plain double buffer)
        if (re != nullptr && atoi(re) >= 2 && atoi(re) < ring) ring = atoi(re);
      }
      const size_t lds = fixed + (size_t)ring * one_stage;
      if (ring >= 2 && fits) {
        p.ext = 1;
        p.team = team;
        p.ring = ring;
        p.loader = loader;
        {
          const char* ro = getenv("NESSAI_AMD_ROUNDS");
          p.rounds_off = (ro != nullptr && ro[0] == '0') ? 1 : 0;
        }
        p.resident = 0;
        p.occupancy = NSA_DEFAULT_OCC;
        p.row_tiles = 1;
        p.skew = 0;
        p.prio = 0;
        long long grid = (n_tiles + tpg - 1) / tpg;
        if (max_groups > 0 && grid > max_groups) grid = max_groups;
        hipError_t e = fn(&p, (unsigned)grid, (unsigned)n_flows, (unsigned)(waves * 64), lds, stream);
        if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "flow kernel launch (team): %s", hipGetErrorString(e));
        g_last_geometry[0] = NSA_DEFAULT_OCC;
        g_last_geometry[3] = 1;
        g_last_geometry[1] = team;
        g_last_geometry[2] = (int)grid;
        g_last_geometry[4] = 0;
        g_last_geometry[5] = team;
        g_last_ring = ring;
        return NSA_OK;
      }
    }
  }
  p.resident = 0;
  g_last_geometry[4] = 0;
  g_last_geometry[5] = 0;
  int occ = force_occ != 0 ? force_occ
                           : nsa_occupancy((int)h[NSA_H_HT], p.precision, n_tiles, n_flows, g_num_cus, waves_per_group);
  if (p.ext && waves_per_group > 8) waves_per_group = 8;
  if (force_occ == 0 && occ == 3 && waves_per_group <= 0 && getenv("NESSAI_AMD_OCC") == nullptr)
    occ = nsa_tuned_occupancy(h, p, max_groups, stream, n_flows, nsa_big_batch_geometry(n_tiles * n_flows, g_num_cus));
  int rt = 1;  // row tiles per wave
  if (occ == NSA_GEO_2X2) {
    rt = (h[NSA_H_HT] <= 2 && p.precision != NSA_PREC_F32) ? 2 : 1;
    occ = 2;
  }
  const size_t stage_bytes = 2 * (size_t)h[NSA_H_MAX_STAGE] * 4;
  auto lds_of = [&](int w) {
    return ((((size_t)w * rt * NSA_ROWS * XS + 3) & ~(size_t)3)) * 4 + (size_t)w * rt * 256 + (size_t)w * 8 + stage_bytes + epi_bytes + prog_bytes;
  };
  auto per_cu_of = [&](int w) {
    long long c = (long long)(g_max_lds / lds_of(w));
    const long long by_waves = (wide ? 4 : 4 * occ) / w;  // occ waves per SIMD (one for the 256-wide instances)
    if (c > by_waves) c = by_waves;
    return c;
  };
  auto groups_of = [&](int w) { return (n_tiles + (long long)w * rt - 1) / ((long long)w * rt); };
  if (occ == 3 && waves_per_group <= 0 && lds_of(12) > g_max_lds) occ = NSA_DEFAULT_OCC;  // tiles too wide
  if (rt == 2 && lds_of(4 * occ) > g_max_lds) {  // tiles too wide for two per wave
    rt = 1;
    occ = NSA_DEFAULT_OCC;
  }
  p.occupancy = occ;
  p.row_tiles = rt;
  {
    // the staged two-tiles geometry (8 waves in lock step through one barrier per stage): the second-dispatched
    // half of the workgroup loses the vector-issue arbitration on every stage; one static s_setprio 1 for it
    // measured 1.565 against 1.581 ms per 1e6 rows of config 3 (the lower half instead: 1.577; gpurun_out r05c2)
    const char* pr = getenv("NESSAI_AMD_PRIO");
    p.prio = pr != nullptr ? atoi(pr) : (rt == 2 ? 1 : 0);
  }
  {
    const char* sk = getenv("NESSAI_AMD_SKEW");  // two-tiles geometry only (see NsaStagePipe::skew)
    p.skew = sk != nullptr ? atoi(sk) : NSA_DEFAULT_SKEW;
  }
  const int nw_max = wide ? 4 : 4 * occ;
  int nw = waves_per_group > 0 ? waves_per_group : (occ == 3 || rt == 2 ? 4 * occ : 2 * occ);
  if (nw > nw_max) nw = nw_max;
  if (waves_per_group <= 0) {
    // small batches: fewer waves per workgroup so that more CUs take part ...
    while (nw > 1 && n_tiles * n_flows < (long long)nw * rt * g_num_cus) nw >>= 1;
    // ... unless that leaves a mostly idle second round which wider workgroups avoid
    while (2 * nw <= nw_max && per_cu_of(2 * nw) >= 1 &&
           groups_of(nw) * n_flows > per_cu_of(nw) * g_num_cus &&
           groups_of(nw) * n_flows <= 2 * per_cu_of(nw) * g_num_cus &&
           groups_of(2 * nw) * n_flows <= per_cu_of(2 * nw) * g_num_cus)
      nw <<= 1;
  }
  while (nw > 1 && lds_of(nw) > g_max_lds) nw >>= 1;
  const size_t lds = lds_of(nw);
  if (lds > g_max_lds)
    return nsa_fail(NSA_E_ARG, "flow needs %zu bytes of LDS per workgroup (> %zu)", lds, g_max_lds);
  const long long n_groups = groups_of(nw);
  long long per_cu = per_cu_of(nw);
  if (per_cu < 1) per_cu = 1;
  long long grid = (per_cu * g_num_cus + n_flows - 1) / n_flows;
  if (max_groups > 0 && grid > max_groups) grid = max_groups;
  if (grid > n_groups) grid = n_groups;
  hipError_t e = fn(&p, (unsigned)grid, (unsigned)n_flows, (unsigned)(nw * 64), lds, stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "flow kernel launch: %s", hipGetErrorString(e));
  g_last_geometry[0] = occ;
  g_last_geometry[3] = rt;
  g_last_geometry[1] = nw;
  g_last_geometry[2] = (int)grid;
  return NSA_OK;
}

// ---------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------

extern "C" int nsa_flow_run(const uint32_t* header, const void* image_dev, int direction, int mode,
                            long long n, const float* x_in_dev, float* x_out_dev, float* ld_out_dev,
                            int* status_dev, int waves_per_group, int max_groups, void* stream) {
  if (image_dev == nullptr || x_in_dev == nullptr || status_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run: NULL image / input / status pointer");
  if (direction != NSA_DIR_FORWARD && direction != NSA_DIR_INVERSE)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run: direction must be 0 or 1");
  if (mode != NSA_MODE_APPLY && mode != NSA_MODE_LOGPROB)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run: mode must be 0 (apply) or 1 (log_prob)");
  if (mode == NSA_MODE_LOGPROB && direction != NSA_DIR_FORWARD)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run: log_prob needs the forward direction");
  NsaLaunch p;
  memset(&p, 0, sizeof(p));
  p.image = (const uint32_t*)image_dev;
  p.direction = direction;
  p.mode = mode;
  p.n = n;
  p.x_in = x_in_dev;
  p.x_out = x_out_dev;
  p.ld_out = ld_out_dev;
  p.status = status_dev;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

extern "C" int nsa_flow_run_ctx(const uint32_t* header, const void* image_dev, int direction, int mode,
                                long long n, const float* x_in_dev, const float* ctx_in_dev, float* x_out_dev,
                                float* ld_out_dev, int* status_dev, int waves_per_group, int max_groups, void* stream) {
  if (image_dev == nullptr || x_in_dev == nullptr || ctx_in_dev == nullptr || status_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_ctx: NULL image / input / context / status pointer");
  if (direction != NSA_DIR_FORWARD && direction != NSA_DIR_INVERSE)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_ctx: direction must be 0 or 1");
  if (mode != NSA_MODE_APPLY && mode != NSA_MODE_LOGPROB)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_ctx: mode must be 0 (apply) or 1 (log_prob)");
  if (mode == NSA_MODE_LOGPROB && direction != NSA_DIR_FORWARD)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_ctx: log_prob needs the forward direction");
  NsaLaunch p;
  memset(&p, 0, sizeof(p));
  p.image = (const uint32_t*)image_dev;
  p.direction = direction;
  p.mode = mode;
  p.n = n;
  p.x_in = x_in_dev;
  p.ctx_in = ctx_in_dev;
  p.x_out = x_out_dev;
  p.ld_out = ld_out_dev;
  p.status = status_dev;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

extern "C" int nsa_flow_run_many(const uint32_t* header, const void* const* image_list_dev, int n_flows,
                                 int direction, int mode, long long n, const float* x_in_dev,
                                 float* x_out_dev, float* ld_out_dev, int* status_dev,
                                 int waves_per_group, int max_groups, void* stream) {
  if (image_list_dev == nullptr || x_in_dev == nullptr || status_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_many: NULL image list / input / status pointer");
  if (n_flows < 1 || n_flows > 65535) return nsa_fail(NSA_E_ARG, "nsa_flow_run_many: n_flows must be in [1, 65535]");
  if (direction != NSA_DIR_FORWARD && direction != NSA_DIR_INVERSE)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_many: direction must be 0 or 1");
  if (mode != NSA_MODE_APPLY && mode != NSA_MODE_LOGPROB)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_many: mode must be 0 (apply) or 1 (log_prob)");
  if (mode == NSA_MODE_LOGPROB && direction != NSA_DIR_FORWARD)
    return nsa_fail(NSA_E_ARG, "nsa_flow_run_many: log_prob needs the forward direction");
  NsaLaunch p;
  memset(&p, 0, sizeof(p));
  p.image_list = (const uint32_t* const*)image_list_dev;
  p.direction = direction;
  p.mode = mode;
  p.n = n;
  p.x_in = x_in_dev;
  p.x_out = x_out_dev;
  p.ld_out = ld_out_dev;
  p.status = status_dev;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream, n_flows);
}

// fills the launch descriptor of the fused draw kernel from the C-ABI arguments
static int nsa_draw_params(NsaLaunch& p, const uint32_t* header, const void* image_dev, long long n,
                           const float* z_in_dev, unsigned long long seed, unsigned long long row_offset,
                           double radius, double latent_temperature, const double* scale_dev,
                           const double* shift_dev, const double* lower_dev, const double* upper_dev,
                           double log_j_rescale, double* x64_out_dev, double* log_q_out_dev, float* z_out_dev,
                           const int* prior_kind_dev, const double* prior_par_dev, double* log_prior_out_dev,
                           const int* rp_kind_dev, const double* rp_par_dev, int* status_dev, bool lean = false) {
  // (lean: no candidate rows - x64_out_dev may be NULL, see NsaLaunch::row_list)
  if (image_dev == nullptr || scale_dev == nullptr || shift_dev == nullptr || lower_dev == nullptr ||
      upper_dev == nullptr || (x64_out_dev == nullptr && !lean) || log_q_out_dev == nullptr || status_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_draw: NULL pointer argument");
  if (header == nullptr) return nsa_fail(NSA_E_ARG, "header is NULL");
  if (!(latent_temperature > 0.0)) return nsa_fail(NSA_E_ARG, "nsa_populate_draw: latent_temperature must be > 0");
  memset(&p, 0, sizeof(p));
  p.image = (const uint32_t*)image_dev;
  p.direction = NSA_DIR_INVERSE;
  p.mode = NSA_MODE_POPULATE;
  p.n = n;
  p.x_in = z_in_dev;
  p.status = status_dev;
  p.seed = seed;
  p.row_offset = row_offset;
  p.radius = radius;
  p.latent_scale = sqrt(latent_temperature);
  p.log_j_latent = (latent_temperature == 1.0) ? 0.0 : (double)header[NSA_H_D] * log(p.latent_scale);
  p.scale = scale_dev;
  p.shift = shift_dev;
  p.lower = lower_dev;
  p.upper = upper_dev;
  p.log_j_rescale = log_j_rescale;
  p.x64_out = x64_out_dev;
  p.log_q_out = log_q_out_dev;
  p.z_out = z_out_dev;
  if (prior_kind_dev != nullptr) {
    if (prior_par_dev == nullptr || log_prior_out_dev == nullptr)
      return nsa_fail(NSA_E_ARG, "nsa_populate_draw: prior parameters / output missing");
    p.prior_kind = prior_kind_dev;
    p.prior_par = prior_par_dev;
    p.log_prior_out = log_prior_out_dev;
  }
  if (rp_kind_dev != nullptr) {
    if (rp_par_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_populate_draw: reparameterisation constants missing");
    p.rp_kind = rp_kind_dev;
    p.rp_par = rp_par_dev;
  }
  return NSA_OK;
}

extern "C" int nsa_populate_draw(const uint32_t* header, const void* image_dev, long long n,
                                 const float* z_in_dev, unsigned long long seed,
                                 unsigned long long row_offset, double radius, double latent_temperature,
                                 const double* scale_dev, const double* shift_dev,
                                 const double* lower_dev, const double* upper_dev,
                                 double log_j_rescale, double* x64_out_dev, double* log_q_out_dev,
                                 float* z_out_dev, const int* prior_kind_dev,
                                 const double* prior_par_dev, double* log_prior_out_dev,
                                 const int* rp_kind_dev, const double* rp_par_dev,
                                 int* status_dev, int waves_per_group, int max_groups, void* stream) {
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, n, z_in_dev, seed, row_offset, radius, latent_temperature,
                           scale_dev, shift_dev, lower_dev, upper_dev, log_j_rescale, x64_out_dev, log_q_out_dev,
                           z_out_dev, prior_kind_dev, prior_par_dev, log_prior_out_dev, rp_kind_dev, rp_par_dev,
                           status_dev, /*lean=*/x64_out_dev == nullptr);
  if (rc != NSA_OK) return rc;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

// The head of one lean pass on its own: draw -> ... -> log_w = log_prior - log_q and the launch's largest weight.
extern "C" int nsa_populate_draw_weights(const uint32_t* header, const void* image_dev, long long n,
                                         unsigned long long seed, unsigned long long row_offset, double radius,
                                         double latent_temperature, const double* scale_dev, const double* shift_dev,
                                         const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                         const int* prior_kind_dev, const double* prior_par_dev,
                                         double* log_prior_out_dev, double log_prior_const, double min_log_q,
                                         const int* rp_kind_dev, const double* rp_par_dev, double* log_w_out_dev,
                                         long long* max_key_dev, int* status_dev, int waves_per_group,
                                         int max_groups, void* stream) {
  if (log_w_out_dev == nullptr || max_key_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_draw_weights: NULL output pointer");
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, n, nullptr, seed, row_offset, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, nullptr, log_w_out_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_out_dev, rp_kind_dev, rp_par_dev, status_dev, true);
  if (rc != NSA_OK) return rc;
  p.log_w_out = log_w_out_dev;
  p.max_key_out = max_key_dev;
  p.log_prior_const = log_prior_const;
  p.min_log_q = min_log_q;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

extern "C" int nsa_importance_draw(const uint32_t* header, const void* image_dev, long long n,
                                   const float* z_in_dev, unsigned long long seed, unsigned long long row_offset,
                                   double unit_eps, double* x64_out_dev, float* xp_out_dev, double* log_j_out_dev,
                                   int* status_dev, int waves_per_group, int max_groups, void* stream) {
  if (header == nullptr || image_dev == nullptr || x64_out_dev == nullptr || xp_out_dev == nullptr ||
      log_j_out_dev == nullptr || status_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_importance_draw: NULL pointer argument");
  if (!(unit_eps >= 0.0 && unit_eps < 0.5)) return nsa_fail(NSA_E_ARG, "nsa_importance_draw: eps must be in [0, 0.5)");
  NsaLaunch p;
  memset(&p, 0, sizeof(p));
  p.image = (const uint32_t*)image_dev;
  p.direction = NSA_DIR_INVERSE;
  p.mode = NSA_MODE_POPULATE;
  p.n = n;
  p.x_in = z_in_dev;
  p.status = status_dev;
  p.seed = seed;
  p.row_offset = row_offset;
  p.radius = INFINITY;
  p.latent_scale = 1.0;
  p.x64_out = x64_out_dev;
  p.log_q_out = log_j_out_dev;
  p.xp_out = xp_out_dev;
  p.unit_eps = unit_eps;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

extern "C" hipError_t nsa_launch_importance_mix(long long, int, const float*, const double*, const double*,
                                                double*, double*, hipStream_t);

extern "C" hipError_t nsa_launch_importance_to_prime(long long, int, const double*, double, float*, double*,
                                                     hipStream_t);

extern "C" int nsa_importance_to_prime(long long n, int dims, const double* x_unit_dev, double eps,
                                       float* x_prime_dev, double* log_j_dev, void* stream) {
  if (n < 0 || dims < 1 || x_unit_dev == nullptr || x_prime_dev == nullptr || log_j_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_importance_to_prime: bad argument");
  if (!(eps >= 0.0 && eps < 0.5)) return nsa_fail(NSA_E_ARG, "nsa_importance_to_prime: eps must be in [0, 0.5)");
  hipError_t e = nsa_launch_importance_to_prime(n, dims, x_unit_dev, eps, x_prime_dev, log_j_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "importance to-prime kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_importance_pack(long long, int, int, const float*, const double*, const double*,
                                                 const double*, int, double*, hipStream_t);

extern "C" int nsa_importance_mix_packed(long long n, int n_flows, int dims, const float* log_prob_all_dev,
                                         const double* log_j_dev, const double* weights_dev, const double* x_unit_dev,
                                         int clip, double* packed_dev, void* stream) {
  if (n < 0 || n_flows < 0 || dims < 1 || log_j_dev == nullptr || weights_dev == nullptr || x_unit_dev == nullptr ||
      packed_dev == nullptr || (n_flows > 0 && log_prob_all_dev == nullptr))
    return nsa_fail(NSA_E_ARG, "nsa_importance_mix_packed: bad argument");
  hipError_t e = nsa_launch_importance_pack(n, n_flows, dims, log_prob_all_dev, log_j_dev, weights_dev, x_unit_dev,
                                            clip, packed_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "importance pack kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_importance_points(long long, int, int, const float*, const double*, const double*,
                                                   const double*, int, const double*, const double*, double, uint32_t*,
                                                   double*, hipStream_t);

extern "C" int nsa_importance_mix_points(long long n, int n_flows, int dims, const float* log_prob_all_dev,
                                         const double* log_j_dev, const double* weights_dev, const double* x_unit_dev,
                                         int clip, const double* lower_dev, const double* upper_dev,
                                         double log_prior_total, void* points_dev, double* aux_dev, void* stream) {
  if (n < 0 || n_flows < 0 || dims < 1 || log_j_dev == nullptr || weights_dev == nullptr || x_unit_dev == nullptr ||
      lower_dev == nullptr || upper_dev == nullptr || points_dev == nullptr || aux_dev == nullptr ||
      (n_flows > 0 && log_prob_all_dev == nullptr))
    return nsa_fail(NSA_E_ARG, "nsa_importance_mix_points: bad argument");
  hipError_t e = nsa_launch_importance_points(n, n_flows, dims, log_prob_all_dev, log_j_dev, weights_dev, x_unit_dev, clip,
                                              lower_dev, upper_dev, log_prior_total, (uint32_t*)points_dev, aux_dev,
                                              (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "importance points kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" int nsa_importance_mix(long long n, int n_flows, const float* log_prob_all_dev, const double* log_j_dev,
                                  const double* weights_dev, double* log_q_all_dev, double* log_Q_dev,
                                  void* stream) {
  if (n < 0 || n_flows < 0 || log_j_dev == nullptr || weights_dev == nullptr || log_q_all_dev == nullptr ||
      log_Q_dev == nullptr || (n_flows > 0 && log_prob_all_dev == nullptr))
    return nsa_fail(NSA_E_ARG, "nsa_importance_mix: bad argument");
  hipError_t e = nsa_launch_importance_mix(n, n_flows, log_prob_all_dev, log_j_dev, weights_dev, log_q_all_dev,
                                           log_Q_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "importance mix kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_weights(long long, const double*, const double*, double, double*,
                                         long long*, hipStream_t, const long long*);
extern "C" hipError_t nsa_launch_post_rules(long long, int, double*, const nsa_post_rules*, const double*,
                                            const long long*, hipStream_t);
extern "C" hipError_t nsa_launch_pass_commit(long long*, long long, const long long*, long long*, hipStream_t);
extern "C" hipError_t nsa_launch_small_tail(long long, int, const double*, const double*, double,
                                            unsigned long long, unsigned long long, const double*, long long,
                                            double*, long long*, hipStream_t);
extern "C" size_t nsa_accept_workspace_bytes(long long);
extern "C" hipError_t nsa_launch_accept(long long, int, const double*, const long long*,
                                        unsigned long long, unsigned long long, const double*,
                                        const double*, long long, double*, long long*, long long*,
                                        void*, hipStream_t, const long long*, long long idx_base = 0);

extern "C" int nsa_populate_weights(long long n, const double* log_q_dev, const double* log_prior_dev,
                                    double log_prior_const, double* log_w_dev, long long* max_key_dev,
                                    void* stream) {
  if (n < 0 || log_q_dev == nullptr || log_w_dev == nullptr || max_key_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_weights: bad argument");
  hipError_t e = nsa_launch_weights(n, log_q_dev, log_prior_dev, log_prior_const, log_w_dev, max_key_dev,
                                    (hipStream_t)stream, nullptr);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "weights kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_truncate(long long, double*, double, const double*, double, hipStream_t);

extern "C" int nsa_populate_truncate(long long n, double* log_q_dev, double min_log_q, const double* log_l_dev,
                                     double log_l_threshold, void* stream) {
  if (n < 0 || log_q_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_populate_truncate: bad argument");
  hipError_t e = nsa_launch_truncate(n, log_q_dev, min_log_q, log_l_dev, log_l_threshold, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "truncate kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" size_t nsa_populate_accept_workspace(long long n) {
  return nsa_accept_workspace_bytes(n < 0 ? 0 : n);
}

extern "C" int nsa_populate_accept(long long n, int D, const double* log_w_dev,
                                   const long long* max_key_dev, unsigned long long seed,
                                   unsigned long long row_offset, const double* u_in_dev,
                                   const double* x64_dev, long long capacity, double* x_acc_dev,
                                   long long* idx_acc_dev, long long* n_accept_dev,
                                   void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (n < 0 || D < 1 || log_w_dev == nullptr || max_key_dev == nullptr || x64_dev == nullptr ||
      x_acc_dev == nullptr || n_accept_dev == nullptr || workspace_dev == nullptr || capacity < 0)
    return nsa_fail(NSA_E_ARG, "nsa_populate_accept: bad argument");
  if (workspace_bytes < nsa_accept_workspace_bytes(n))
    return nsa_fail(NSA_E_ARG, "nsa_populate_accept: workspace too small (%zu < %zu)", workspace_bytes,
                    nsa_accept_workspace_bytes(n));
  hipError_t e = nsa_launch_accept(n, D, log_w_dev, max_key_dev, seed, row_offset, u_in_dev, x64_dev,
                                   capacity, x_acc_dev, idx_acc_dev, n_accept_dev, workspace_dev,
                                   (hipStream_t)stream, nullptr);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "accept kernels: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_log_likelihood(int, long long, int, const double*, const double*, const double*,
                                                double, double*, hipStream_t);

extern "C" int nsa_log_likelihood(int kind, long long n, int D, const double* x64_dev, const double* mu_dev,
                                  const double* sigma_dev, double constant, double* out_dev, void* stream) {
  if (kind < 0 || kind > 1 || n < 0 || D < 2 || D > 129 || x64_dev == nullptr || out_dev == nullptr ||
      ((mu_dev == nullptr) != (sigma_dev == nullptr)))
    return nsa_fail(NSA_E_ARG, "nsa_log_likelihood: bad argument (kind 0/1, 2 <= D <= 129)");
  hipError_t e = nsa_launch_log_likelihood(kind, n, D, x64_dev, mu_dev, sigma_dev, constant, out_dev,
                                           (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "log-likelihood kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

// A train of populate passes without host round trips (see include/nessai_amd.h).  x64_dev == NULL: the lean
// form - no candidate rows are written, the accepted rows are listed by their global Philox rows in idx_acc_dev.
static int nsa_passes_impl(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                           unsigned long long seed, unsigned long long row_offset, double radius,
                           double latent_temperature, const double* scale_dev, const double* shift_dev,
                           const double* lower_dev, const double* upper_dev, double log_j_rescale,
                           double* x64_dev, double* log_q_dev, const int* prior_kind_dev,
                           const double* prior_par_dev, double* log_prior_dev, double log_prior_const,
                           const int* rp_kind_dev, const double* rp_par_dev, double* log_w_dev,
                           long long* max_keys_dev, long long capacity, double* x_acc_dev, long long* idx_acc_dev,
                           long long* state_dev, void* workspace_dev, size_t workspace_bytes,
                           int* status_dev, int waves_per_group, int max_groups, void* stream,
                           const nsa_post_rules* rules, const char* who) {
  const bool lean = x64_dev == nullptr;
  if (n <= 0 || n_passes < 1 || log_w_dev == nullptr || max_keys_dev == nullptr ||
      (lean ? idx_acc_dev == nullptr : x_acc_dev == nullptr) || state_dev == nullptr || workspace_dev == nullptr ||
      capacity < 0)
    return nsa_fail(NSA_E_ARG, "%s: bad argument", who);
  if (workspace_bytes < nsa_accept_workspace_bytes(n)) return nsa_fail(NSA_E_ARG, "%s: workspace too small", who);
  if (lean && rules != nullptr && rules->ll_kind >= 0)
    return nsa_fail(NSA_E_ARG, "%s: the likelihood rule needs the candidate rows (use nsa_populate_passes)", who);
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, n, nullptr, seed, row_offset, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, x64_dev, log_q_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_dev, rp_kind_dev, rp_par_dev, status_dev, lean);
  if (rc != NSA_OK) return rc;
  p.stop = state_dev;
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)header[NSA_H_D];
  const double* per_row_prior = (prior_kind_dev != nullptr) ? log_prior_dev : nullptr;
  for (int i = 0; i < n_passes; ++i) {
    const unsigned long long off = row_offset + (unsigned long long)i * (unsigned long long)n;
    p.row_offset = off;
    if (lean) {  // weights and their maximum come out of the draw kernel itself (NsaLaunch::log_w_out)
      p.log_w_out = log_w_dev;
      p.max_key_out = max_keys_dev + i;
      p.log_prior_const = log_prior_const;
      p.min_log_q = rules != nullptr ? rules->min_log_q : -INFINITY;
    }
    rc = nsa_launch_flow(header, p, waves_per_group, max_groups, s);
    if (rc != NSA_OK) return rc;
    if (rules != nullptr && !lean) {
      hipError_t er = nsa_launch_post_rules(n, D, log_q_dev, rules, x64_dev, state_dev, s);
      if (er != hipSuccess) return nsa_fail(NSA_E_HIP, "populate pass %d (rules after the flow): %s", i, hipGetErrorString(er));
    }
    if (n <= 16384 && !lean) {  // NSA_SMALL_N: the whole tail of the pass in one single-workgroup kernel
      hipError_t es = nsa_launch_small_tail(n, D, log_q_dev, per_row_prior, log_prior_const, seed, off, x64_dev,
                                            capacity, x_acc_dev, state_dev, s);
      if (es != hipSuccess) return nsa_fail(NSA_E_HIP, "populate pass %d: %s", i, hipGetErrorString(es));
      continue;
    }
    hipError_t e = hipSuccess;
    if (!lean)
      e = nsa_launch_weights(n, log_q_dev, per_row_prior, log_prior_const, log_w_dev, max_keys_dev + i, s, state_dev);
    if (e == hipSuccess)
      e = nsa_launch_accept(n, D, log_w_dev, max_keys_dev + i, seed, off, nullptr, x64_dev, capacity, x_acc_dev,
                            idx_acc_dev, state_dev + 4, workspace_dev, s, state_dev, (long long)off);
    if (e == hipSuccess) e = nsa_launch_pass_commit(state_dev, capacity, nullptr, nullptr, s);
    if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "populate pass %d: %s", i, hipGetErrorString(e));
  }
  return NSA_OK;
}

extern "C" int nsa_populate_passes(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                   unsigned long long seed, unsigned long long row_offset, double radius,
                                   double latent_temperature, const double* scale_dev, const double* shift_dev,
                                   const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                   double* x64_dev, double* log_q_dev, const int* prior_kind_dev,
                                   const double* prior_par_dev, double* log_prior_dev, double log_prior_const,
                                   const int* rp_kind_dev, const double* rp_par_dev, double* log_w_dev,
                                   long long* max_keys_dev, long long capacity, double* x_acc_dev,
                                   long long* state_dev, void* workspace_dev, size_t workspace_bytes,
                                   int* status_dev, int waves_per_group, int max_groups, void* stream,
                                   const nsa_post_rules* rules) {
  if (x64_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_populate_passes: bad argument");
  return nsa_passes_impl(header, image_dev, n, n_passes, seed, row_offset, radius, latent_temperature, scale_dev,
                         shift_dev, lower_dev, upper_dev, log_j_rescale, x64_dev, log_q_dev, prior_kind_dev,
                         prior_par_dev, log_prior_dev, log_prior_const, rp_kind_dev, rp_par_dev, log_w_dev, max_keys_dev,
                         capacity, x_acc_dev, nullptr, state_dev, workspace_dev, workspace_bytes, status_dev,
                         waves_per_group, max_groups, stream, rules, "nsa_populate_passes");
}

extern "C" int nsa_populate_passes_lean(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                        unsigned long long seed, unsigned long long row_offset, double radius,
                                        double latent_temperature, const double* scale_dev, const double* shift_dev,
                                        const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                        double* log_q_dev, const int* prior_kind_dev, const double* prior_par_dev,
                                        double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                                        const double* rp_par_dev, double* log_w_dev, long long* max_keys_dev,
                                        long long capacity, long long* idx_acc_dev, long long* state_dev,
                                        void* workspace_dev, size_t workspace_bytes, int* status_dev,
                                        int waves_per_group, int max_groups, void* stream,
                                        const nsa_post_rules* rules) {
  return nsa_passes_impl(header, image_dev, n, n_passes, seed, row_offset, radius, latent_temperature, scale_dev,
                         shift_dev, lower_dev, upper_dev, log_j_rescale, nullptr, log_q_dev, prior_kind_dev,
                         prior_par_dev, log_prior_dev, log_prior_const, rp_kind_dev, rp_par_dev, log_w_dev, max_keys_dev,
                         capacity, nullptr, idx_acc_dev, state_dev, workspace_dev, workspace_bytes, status_dev,
                         waves_per_group, max_groups, stream, rules, "nsa_populate_passes_lean");
}

// The accepted rows of lean passes, computed again from their global Philox rows (see include/nessai_amd.h).
extern "C" int nsa_populate_replay(const uint32_t* header, const void* image_dev, long long m,
                                   const long long* row_list_dev, unsigned long long seed, double radius,
                                   double latent_temperature, const double* scale_dev, const double* shift_dev,
                                   const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                   double* x64_out_dev, double* log_q_out_dev, const int* prior_kind_dev,
                                   const double* prior_par_dev, double* log_prior_out_dev, const int* rp_kind_dev,
                                   const double* rp_par_dev, int* status_dev, int waves_per_group, int max_groups,
                                   void* stream) {
  if (m < 0 || row_list_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_populate_replay: bad argument");
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, m, nullptr, seed, 0ull, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, x64_out_dev, log_q_out_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_out_dev, rp_kind_dev, rp_par_dev, status_dev);
  if (rc != NSA_OK) return rc;
  p.row_list = row_list_dev;
  return nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
}

// The pieces of one pass for SHARDED trains (the collectives of torch.distributed go in between:
// head -> all-reduce(max) of max_key -> tail -> all-reduce(sum) of state[4] -> commit).
extern "C" int nsa_populate_pass_head(const uint32_t* header, const void* image_dev, long long n,
                                      unsigned long long seed, unsigned long long row_offset, double radius,
                                      double latent_temperature, const double* scale_dev, const double* shift_dev,
                                      const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                      double* x64_dev, double* log_q_dev, const int* prior_kind_dev,
                                      const double* prior_par_dev, double* log_prior_dev, double log_prior_const,
                                      const int* rp_kind_dev, const double* rp_par_dev, double* log_w_dev,
                                      long long* max_key_dev, const long long* state_dev, int* status_dev,
                                      int waves_per_group, int max_groups, void* stream) {
  if (n <= 0 || log_w_dev == nullptr || max_key_dev == nullptr || state_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_pass_head: bad argument");
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, n, nullptr, seed, row_offset, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, x64_dev, log_q_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_dev, rp_kind_dev, rp_par_dev, status_dev);
  if (rc != NSA_OK) return rc;
  p.stop = state_dev;
  rc = nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
  if (rc != NSA_OK) return rc;
  hipError_t e = nsa_launch_weights(n, log_q_dev, (prior_kind_dev != nullptr) ? log_prior_dev : nullptr,
                                    log_prior_const, log_w_dev, max_key_dev, (hipStream_t)stream, state_dev);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "weights kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" int nsa_populate_pass_tail(long long n, int D, const double* log_w_dev, const long long* max_key_dev,
                                      unsigned long long seed, unsigned long long row_offset,
                                      const double* x64_dev, long long capacity, double* x_acc_dev,
                                      long long* state_dev, void* workspace_dev, size_t workspace_bytes,
                                      void* stream) {
  if (n <= 0 || D < 1 || log_w_dev == nullptr || max_key_dev == nullptr || x64_dev == nullptr ||
      x_acc_dev == nullptr || state_dev == nullptr || workspace_dev == nullptr || capacity < 0 ||
      workspace_bytes < nsa_accept_workspace_bytes(n))
    return nsa_fail(NSA_E_ARG, "nsa_populate_pass_tail: bad argument");
  hipError_t e = nsa_launch_accept(n, D, log_w_dev, max_key_dev, seed, row_offset, nullptr, x64_dev, capacity,
                                   x_acc_dev, nullptr, state_dev + 4, workspace_dev, (hipStream_t)stream, state_dev);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "accept kernels: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" int nsa_populate_pass_commit(long long* state_dev, long long capacity, const long long* global_count_dev,
                                        long long* history_dev, void* stream) {
  if (state_dev == nullptr || capacity < 0) return nsa_fail(NSA_E_ARG, "nsa_populate_pass_commit: bad argument");
  hipError_t e = nsa_launch_pass_commit(state_dev, capacity, global_count_dev, history_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "commit kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

// Trains of sharded passes: two collectives per train (see include/nessai_amd.h).
extern "C" hipError_t nsa_launch_train_scratch(const long long*, long long*, hipStream_t);
extern "C" hipError_t nsa_launch_train_commit(long long*, long long, int, const long long*, const long long*,
                                              long long*, hipStream_t);

static int nsa_train_heads_impl(bool lean, const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                        unsigned long long seed, unsigned long long row_offset,
                                        unsigned long long row_stride, double radius, double latent_temperature,
                                        const double* scale_dev, const double* shift_dev, const double* lower_dev,
                                        const double* upper_dev, double log_j_rescale, double* x64_many_dev,
                                        double* log_q_dev, const int* prior_kind_dev, const double* prior_par_dev,
                                        double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                                        const double* rp_par_dev, double* log_w_many_dev, long long* max_keys_dev,
                                        const long long* state_dev, int* status_dev, int waves_per_group,
                                        int max_groups, void* stream, const nsa_post_rules* rules) {
  if (n <= 0 || n_passes < 1 || (x64_many_dev == nullptr) != lean || log_w_many_dev == nullptr ||
      max_keys_dev == nullptr || state_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_train_heads: bad argument");
  if (lean && rules != nullptr && rules->ll_kind >= 0)
    return nsa_fail(NSA_E_ARG, "nsa_populate_train_heads_lean: the likelihood rule needs the candidate rows");
  NsaLaunch p;
  int rc = nsa_draw_params(p, header, image_dev, n, nullptr, seed, row_offset, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, x64_many_dev, log_q_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_dev, rp_kind_dev, rp_par_dev, status_dev, lean);
  if (rc != NSA_OK) return rc;
  p.stop = state_dev;
  const long long D = (long long)header[NSA_H_D];
  const double* per_row_prior = (prior_kind_dev != nullptr) ? log_prior_dev : nullptr;
  for (int i = 0; i < n_passes; ++i) {
    p.row_offset = row_offset + (unsigned long long)i * row_stride;
    p.x64_out = lean ? nullptr : x64_many_dev + (long long)i * n * D;
    if (lean) {  // weights and their maximum come out of the draw kernel itself
      p.log_w_out = log_w_many_dev + (long long)i * n;
      p.max_key_out = max_keys_dev + i;
      p.log_prior_const = log_prior_const;
      p.min_log_q = rules != nullptr ? rules->min_log_q : -INFINITY;
      rc = nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
      if (rc != NSA_OK) return rc;
      continue;
    }
    rc = nsa_launch_flow(header, p, waves_per_group, max_groups, (hipStream_t)stream);
    if (rc != NSA_OK) return rc;
    if (rules != nullptr) {
      hipError_t er = nsa_launch_post_rules(n, (int)D, log_q_dev, rules, p.x64_out, state_dev, (hipStream_t)stream);
      if (er != hipSuccess) return nsa_fail(NSA_E_HIP, "train head %d (rules after the flow): %s", i, hipGetErrorString(er));
    }
    hipError_t e = nsa_launch_weights(n, log_q_dev, per_row_prior, log_prior_const, log_w_many_dev + (long long)i * n,
                                      max_keys_dev + i, (hipStream_t)stream, state_dev);
    if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "train head %d: %s", i, hipGetErrorString(e));
  }
  return NSA_OK;
}

extern "C" int nsa_populate_train_heads(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                        unsigned long long seed, unsigned long long row_offset,
                                        unsigned long long row_stride, double radius, double latent_temperature,
                                        const double* scale_dev, const double* shift_dev, const double* lower_dev,
                                        const double* upper_dev, double log_j_rescale, double* x64_many_dev,
                                        double* log_q_dev, const int* prior_kind_dev, const double* prior_par_dev,
                                        double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                                        const double* rp_par_dev, double* log_w_many_dev, long long* max_keys_dev,
                                        const long long* state_dev, int* status_dev, int waves_per_group,
                                        int max_groups, void* stream, const nsa_post_rules* rules) {
  return nsa_train_heads_impl(false, header, image_dev, n, n_passes, seed, row_offset, row_stride, radius,
                              latent_temperature, scale_dev, shift_dev, lower_dev, upper_dev, log_j_rescale, x64_many_dev,
                              log_q_dev, prior_kind_dev, prior_par_dev, log_prior_dev, log_prior_const, rp_kind_dev,
                              rp_par_dev, log_w_many_dev, max_keys_dev, state_dev, status_dev, waves_per_group, max_groups,
                              stream, rules);
}

extern "C" int nsa_populate_train_heads_lean(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                             unsigned long long seed, unsigned long long row_offset,
                                             unsigned long long row_stride, double radius, double latent_temperature,
                                             const double* scale_dev, const double* shift_dev, const double* lower_dev,
                                             const double* upper_dev, double log_j_rescale, double* log_q_dev,
                                             const int* prior_kind_dev, const double* prior_par_dev,
                                             double* log_prior_dev, double log_prior_const, const int* rp_kind_dev,
                                             const double* rp_par_dev, double* log_w_many_dev, long long* max_keys_dev,
                                             const long long* state_dev, int* status_dev, int waves_per_group,
                                             int max_groups, void* stream, const nsa_post_rules* rules) {
  return nsa_train_heads_impl(true, header, image_dev, n, n_passes, seed, row_offset, row_stride, radius,
                              latent_temperature, scale_dev, shift_dev, lower_dev, upper_dev, log_j_rescale, nullptr,
                              log_q_dev, prior_kind_dev, prior_par_dev, log_prior_dev, log_prior_const, rp_kind_dev,
                              rp_par_dev, log_w_many_dev, max_keys_dev, state_dev, status_dev, waves_per_group, max_groups,
                              stream, rules);
}

static int nsa_train_tails_impl(long long n, int D, int n_passes, const double* log_w_many_dev,
                                        const long long* max_keys_dev, unsigned long long seed,
                                        unsigned long long row_offset, unsigned long long row_stride,
                                        const double* x64_many_dev, long long capacity, double* x_acc_dev, long long* idx_acc_dev,
                                        const long long* state_dev, long long* scratch_dev,
                                        long long* local_counts_dev, void* workspace_dev, size_t workspace_bytes,
                                        void* stream) {
  if (n <= 0 || D < 1 || n_passes < 1 || log_w_many_dev == nullptr || max_keys_dev == nullptr ||
      (x64_many_dev == nullptr ? idx_acc_dev == nullptr : x_acc_dev == nullptr) || state_dev == nullptr || scratch_dev == nullptr ||
      local_counts_dev == nullptr || workspace_dev == nullptr || capacity < 0 ||
      workspace_bytes < nsa_accept_workspace_bytes(n))
    return nsa_fail(NSA_E_ARG, "nsa_populate_train_tails: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = nsa_launch_train_scratch(state_dev, scratch_dev, s);
  for (int i = 0; i < n_passes && e == hipSuccess; ++i) {
    e = nsa_launch_accept(n, D, log_w_many_dev + (long long)i * n, max_keys_dev + i, seed,
                          row_offset + (unsigned long long)i * row_stride, nullptr,
                          x64_many_dev != nullptr ? x64_many_dev + (long long)i * n * D : nullptr, capacity, x_acc_dev,
                          idx_acc_dev, scratch_dev + 4, workspace_dev, s, scratch_dev,
                          (long long)(row_offset + (unsigned long long)i * row_stride));
    // (scratch[1] counts the passes of this train: history index = pass within the train)
    if (e == hipSuccess) e = nsa_launch_pass_commit(scratch_dev, capacity, nullptr, local_counts_dev, s);
  }
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "train tails: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" int nsa_populate_train_tails(long long n, int D, int n_passes, const double* log_w_many_dev,
                                        const long long* max_keys_dev, unsigned long long seed,
                                        unsigned long long row_offset, unsigned long long row_stride,
                                        const double* x64_many_dev, long long capacity, double* x_acc_dev,
                                        const long long* state_dev, long long* scratch_dev,
                                        long long* local_counts_dev, void* workspace_dev, size_t workspace_bytes,
                                        void* stream) {
  if (x64_many_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_populate_train_tails: bad argument");
  return nsa_train_tails_impl(n, D, n_passes, log_w_many_dev, max_keys_dev, seed, row_offset, row_stride, x64_many_dev,
                              capacity, x_acc_dev, nullptr, state_dev, scratch_dev, local_counts_dev, workspace_dev,
                              workspace_bytes, stream);
}

extern "C" int nsa_populate_train_tails_lean(long long n, int D, int n_passes, const double* log_w_many_dev,
                                             const long long* max_keys_dev, unsigned long long seed,
                                             unsigned long long row_offset, unsigned long long row_stride,
                                             long long capacity, long long* idx_acc_dev, const long long* state_dev,
                                             long long* scratch_dev, long long* local_counts_dev, void* workspace_dev,
                                             size_t workspace_bytes, void* stream) {
  return nsa_train_tails_impl(n, D, n_passes, log_w_many_dev, max_keys_dev, seed, row_offset, row_stride, nullptr,
                              capacity, nullptr, idx_acc_dev, state_dev, scratch_dev, local_counts_dev, workspace_dev,
                              workspace_bytes, stream);
}

extern "C" int nsa_populate_train_commit(long long* state_dev, long long capacity, int n_passes,
                                         const long long* local_counts_dev, const long long* global_counts_dev,
                                         long long* history_dev, void* stream) {
  if (state_dev == nullptr || capacity < 0 || n_passes < 1 || local_counts_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_train_commit: bad argument");
  hipError_t e = nsa_launch_train_commit(state_dev, capacity, n_passes, local_counts_dev, global_counts_dev,
                                         history_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "train commit kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_sumexp(long long, int, const double*, const long long*, double*, hipStream_t);

extern "C" int nsa_populate_sumexp(long long n, int n_passes, const double* log_w_many_dev,
                                   const long long* max_keys_dev, double* out_dev, void* stream) {
  if (n <= 0 || n_passes < 1 || log_w_many_dev == nullptr || max_keys_dev == nullptr || out_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_populate_sumexp: bad argument");
  hipError_t e = nsa_launch_sumexp(n, n_passes, log_w_many_dev, max_keys_dev, out_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "sumexp kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

extern "C" hipError_t nsa_launch_repack(long long, const int*, const int*, const int*, const unsigned char*,
                                        const float*, int, uint32_t*, hipStream_t);

extern "C" int nsa_repack_image(long long n_words, const int* pos_dev, const int* e0_dev, const int* e1_dev,
                                const unsigned char* kind_dev, const float* params_dev, int precision,
                                void* image_dev, void* stream) {
  if (n_words < 0 || (n_words > 0 && (pos_dev == nullptr || e0_dev == nullptr || e1_dev == nullptr ||
                                      kind_dev == nullptr || params_dev == nullptr || image_dev == nullptr)))
    return nsa_fail(NSA_E_ARG, "nsa_repack_image: bad argument");
  if (precision != NSA_PREC_F32 && precision != NSA_PREC_BF16X3 && precision != NSA_PREC_F16X3)
    return nsa_fail(NSA_E_ARG, "nsa_repack_image: unknown precision code %d", precision);
  hipError_t e = nsa_launch_repack(n_words, pos_dev, e0_dev, e1_dev, kind_dev, params_dev, precision,
                                   (uint32_t*)image_dev, (hipStream_t)stream);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "repack kernel: %s", hipGetErrorString(e));
  return NSA_OK;
}

// A train of SMALL passes as one draw launch + two tail launches (see include/nessai_amd.h).
extern "C" hipError_t nsa_launch_probe_compare(long long, int, const float*, const float*, const float*, const float*,
                                               const float*, unsigned long long*, hipStream_t);
extern "C" hipError_t nsa_launch_probe_nudge(long long, const float*, float*, hipStream_t);
extern "C" hipError_t nsa_launch_probe_range(long long, const float*, float, int*, hipStream_t);

extern "C" int nsa_probe_compare(long long n, int D, const float* z_dev, const float* ld_dev, const float* lp_dev,
                                 const float* ld_ref_dev, const float* lp_ref_dev, double* out_dev, void* stream) {
  if (n < 0 || D < 1 || z_dev == nullptr || ld_dev == nullptr || lp_dev == nullptr || ld_ref_dev == nullptr ||
      lp_ref_dev == nullptr || out_dev == nullptr)
    return nsa_fail(NSA_E_ARG, "nsa_probe_compare: bad argument");
  if (n == 0) return NSA_OK;
  hipError_t e = nsa_launch_probe_compare(n, D, z_dev, ld_dev, lp_dev, ld_ref_dev, lp_ref_dev,
                                          (unsigned long long*)out_dev, (hipStream_t)stream);
  return e == hipSuccess ? NSA_OK : nsa_fail(NSA_E_HIP, "nsa_probe_compare: %s", hipGetErrorString(e));
}

extern "C" int nsa_probe_nudge(long long n, const float* x_in_dev, float* x_out_dev, void* stream) {
  if (n < 0 || x_in_dev == nullptr || x_out_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_probe_nudge: bad argument");
  if (n == 0) return NSA_OK;
  hipError_t e = nsa_launch_probe_nudge(n, x_in_dev, x_out_dev, (hipStream_t)stream);
  return e == hipSuccess ? NSA_OK : nsa_fail(NSA_E_HIP, "nsa_probe_nudge: %s", hipGetErrorString(e));
}

extern "C" int nsa_probe_range(long long n, const float* values_dev, float limit, int* flag_dev, void* stream) {
  if (n < 0 || values_dev == nullptr || flag_dev == nullptr) return nsa_fail(NSA_E_ARG, "nsa_probe_range: bad argument");
  if (n == 0) return NSA_OK;
  hipError_t e = nsa_launch_probe_range(n, values_dev, limit, flag_dev, (hipStream_t)stream);
  return e == hipSuccess ? NSA_OK : nsa_fail(NSA_E_HIP, "nsa_probe_range: %s", hipGetErrorString(e));
}

extern "C" hipError_t nsa_launch_small_train(long long, int, int, const double*, const double*, double,
                                             unsigned long long, unsigned long long, const double*, long long, double*,
                                             long long*, long long, double*, long long*, hipStream_t);

extern "C" int nsa_populate_passes_small(const uint32_t* header, const void* image_dev, long long n, int n_passes,
                                         unsigned long long seed, unsigned long long row_offset, double radius,
                                         double latent_temperature, const double* scale_dev, const double* shift_dev,
                                         const double* lower_dev, const double* upper_dev, double log_j_rescale,
                                         double* x64_many_dev, double* log_q_many_dev, const int* prior_kind_dev,
                                         const double* prior_par_dev, double* log_prior_many_dev,
                                         double log_prior_const, const int* rp_kind_dev, const double* rp_par_dev,
                                         long long stage_rows, double* stage_dev, long long* counts_dev,
                                         long long capacity, double* x_acc_dev, long long* state_dev, int* status_dev,
                                         int waves_per_group, int max_groups, void* stream,
                                         const nsa_post_rules* rules) {
  if (n <= 0 || n > 16384 || n_passes < 1 || n_passes > 64 || x64_many_dev == nullptr || log_q_many_dev == nullptr ||
      stage_dev == nullptr || counts_dev == nullptr || x_acc_dev == nullptr || state_dev == nullptr || capacity < 0 ||
      stage_rows < 1)
    return nsa_fail(NSA_E_ARG, "nsa_populate_passes_small: bad argument (1 <= n <= 16384, 1 <= n_passes <= 64)");
  NsaLaunch p;
  const long long rows = n * (long long)n_passes;
  int rc = nsa_draw_params(p, header, image_dev, rows, nullptr, seed, row_offset, radius, latent_temperature, scale_dev,
                           shift_dev, lower_dev, upper_dev, log_j_rescale, x64_many_dev, log_q_many_dev, nullptr,
                           prior_kind_dev, prior_par_dev, log_prior_many_dev, rp_kind_dev, rp_par_dev, status_dev);
  if (rc != NSA_OK) return rc;
  p.stop = state_dev;
  hipStream_t s = (hipStream_t)stream;
  rc = nsa_launch_flow(header, p, waves_per_group, max_groups, s);
  if (rc != NSA_OK) return rc;
  const int D = (int)header[NSA_H_D];
  if (rules != nullptr) {
    hipError_t er = nsa_launch_post_rules(rows, D, log_q_many_dev, rules, x64_many_dev, state_dev, s);
    if (er != hipSuccess) return nsa_fail(NSA_E_HIP, "small train (rules after the flow): %s", hipGetErrorString(er));
  }
  hipError_t e = nsa_launch_small_train(n, D, n_passes, log_q_many_dev,
                                        (prior_kind_dev != nullptr) ? log_prior_many_dev : nullptr, log_prior_const,
                                        seed, row_offset, x64_many_dev, stage_rows, stage_dev, counts_dev, capacity,
                                        x_acc_dev, state_dev, s);
  if (e != hipSuccess) return nsa_fail(NSA_E_HIP, "small train: %s", hipGetErrorString(e));
  return NSA_OK;
}
