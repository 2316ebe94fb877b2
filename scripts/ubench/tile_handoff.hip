// Micro-benchmark (diagnostic, not part of the library): what it costs to hand 32-row tiles from one CU to
// another through L2 - the step a "couplings resident per CU, rows travel" redesign of the config-3 draw kernel
// would add six times per tile.  Pairs of workgroups (producer p, consumer p + 1): the
// producer writes a tile of TILE_BYTES into slot (i mod DEPTH) of a ring in global memory and publishes i + 1 in a
// flag (release); the consumer spins on the flag (acquire), reads the tile, and returns the slot through a second
// flag.  Prints microseconds per tile per pair for ring depths 1 / 2 / 4 / 8, with the two workgroups on the same
// XCD (block ids 8 apart: ids are dealt round-robin over the 8 XCDs) and on neighbouring XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/tile_handoff scripts/ubench/tile_handoff.hip && scripts/ubench/tile_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int TILE_FLOATS, int DEPTH>
__global__ void __launch_bounds__(256) handoff(float* ring, unsigned* flags, int tiles, int partner_stride, float* sink) {
  // blocks 2 k ... : block b < half is a producer, b + partner offset its consumer
  const int pairs = gridDim.x / 2;
  const int b = blockIdx.x;
  const bool producer = (b % (2 * partner_stride)) < partner_stride;
  const int pair = (b / (2 * partner_stride)) * partner_stride + (b % partner_stride);
  if (pair >= pairs) return;
  float* my_ring = ring + (size_t)pair * DEPTH * TILE_FLOATS;
  unsigned* full = flags + pair * 64;       // tiles published
  unsigned* freed = flags + pair * 64 + 32;  // tiles consumed
  float acc = 0.0f;
  for (int i = 0; i < tiles; ++i) {
    float* slot = my_ring + (size_t)(i % DEPTH) * TILE_FLOATS;
    if (producer) {
      if (threadIdx.x == 0)
        while ((int)__hip_atomic_load(freed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) + DEPTH <= i) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      for (int k = threadIdx.x; k < TILE_FLOATS; k += 256) __builtin_nontemporal_store((float)(i + k), slot + k);
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(full, (unsigned)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0)
        while ((int)__hip_atomic_load(full, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= i) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      for (int k = threadIdx.x; k < TILE_FLOATS; k += 256) acc += __builtin_nontemporal_load(slot + k);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(freed, (unsigned)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!producer) sink[b * 256 + threadIdx.x] = acc;
}

template <int TILE_FLOATS, int DEPTH>
static double run(int partner_stride, int tiles) {
  const int blocks = 256;  // one workgroup per CU: 128 pairs
  float *ring, *sink;
  unsigned* flags;
  hipMalloc(&ring, sizeof(float) * (size_t)(blocks / 2) * DEPTH * TILE_FLOATS);
  hipMalloc(&sink, sizeof(float) * blocks * 256);
  hipMalloc(&flags, sizeof(unsigned) * (blocks / 2) * 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float ms = 0.0f;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(flags, 0, sizeof(unsigned) * (blocks / 2) * 64);
    hipEventRecord(a);
    handoff<TILE_FLOATS, DEPTH><<<blocks, 256>>>(ring, flags, tiles, partner_stride, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  hipFree(ring);
  hipFree(sink);
  hipFree(flags);
  return ms * 1e3 / tiles;  // us per tile per pair (all 128 pairs run concurrently)
}

int main() {
  const int tiles = 4000;
  printf("us per tile handed over, per producer/consumer pair (128 pairs at once, one workgroup per CU)\n");
  printf("%-34s %8s %8s %8s %8s\n", "tile / placement", "depth 1", "depth 2", "depth 4", "depth 8");
  printf("%-34s %8.2f %8.2f %8.2f %8.2f\n", "4 KiB (x tile), same XCD", run<1024, 1>(8, tiles), run<1024, 2>(8, tiles),
         run<1024, 4>(8, tiles), run<1024, 8>(8, tiles));
  printf("%-34s %8.2f %8.2f %8.2f %8.2f\n", "4 KiB (x tile), next XCD", run<1024, 1>(1, tiles), run<1024, 2>(1, tiles),
         run<1024, 4>(1, tiles), run<1024, 8>(1, tiles));
  printf("%-34s %8.2f %8.2f %8.2f %8.2f\n", "12 KiB (x + hidden state), same XCD", run<3072, 1>(8, tiles), run<3072, 2>(8, tiles),
         run<3072, 4>(8, tiles), run<3072, 8>(8, tiles));
  printf("%-34s %8.2f %8.2f %8.2f %8.2f\n", "12 KiB (x + hidden state), next XCD", run<3072, 1>(1, tiles), run<3072, 2>(1, tiles),
         run<3072, 4>(1, tiles), run<3072, 8>(1, tiles));
  return 0;
}
