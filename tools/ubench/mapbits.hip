// Micro-benchmark of the occupancy-image sweep (k_mapbits): variants A/B'd in ONE process, interleaved rounds
// (cdna_hip_programming.md rule 24).  Build: hipcc --offload-arch=gfx950 -O3 -o mapbits mapbits.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int UNROLL, bool NT, bool SHUF, bool SPREAD = false, bool DPP = false>
__global__ __launch_bounds__(256) void v_f4(const float* __restrict__ map, uint64_t n, float thr, unsigned long long* __restrict__ bits, unsigned long long* count)
{
  const uint64_t n4 = n >> 2;
  const uint64_t n_words = (n + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
  const float4* map4 = reinterpret_cast<const float4*>(map);
  unsigned long long local = 0;
  const uint64_t n_groups_round = (((n + 3) >> 2) + 63) & ~63ull;
  for (uint64_t g0 = tid; g0 < n_groups_round; g0 += nthreads * UNROLL)
  {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      const uint64_t gi = g0 + (uint64_t)u * nthreads;
      if (gi < n4) { if (NT) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(map4) + gi); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = map4[gi]; }
      else v[u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      const uint64_t gi = g0 + (uint64_t)u * nthreads;
      const unsigned nib = (v[u].x > thr ? 1u : 0u) | (v[u].y > thr ? 2u : 0u) | (v[u].z > thr ? 4u : 0u) | (v[u].w > thr ? 8u : 0u);
      unsigned long long w = (unsigned long long)nib << ((lane & 15u) * 4u);
      if (DPP)
      {
        // OR-reduce the 16 lanes of a DPP row into its lane 0: row_shl:8,4,2,1 with bound_ctrl (out-of-row reads give 0)
        unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
        lo |= __builtin_amdgcn_update_dpp(0, lo, 0x108, 0xf, 0xf, true); hi |= __builtin_amdgcn_update_dpp(0, hi, 0x108, 0xf, 0xf, true);
        lo |= __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0xf, true); hi |= __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0xf, true);
        lo |= __builtin_amdgcn_update_dpp(0, lo, 0x102, 0xf, 0xf, true); hi |= __builtin_amdgcn_update_dpp(0, hi, 0x102, 0xf, 0xf, true);
        lo |= __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true); hi |= __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
        w = ((unsigned long long)hi << 32) | lo;
      }
      else if (SHUF) { w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4); w |= __shfl_xor(w, 8); }
      const uint64_t wi = gi >> 4;
      if ((lane & 15u) == 0 && wi < n_words) { bits[wi] = w; local += __popcll(w); }
    }
  }
  for (int s = 32; s > 0; s >>= 1) local += __shfl_xor(local, s);
  if (!SPREAD) { if (lane == 0 && local) atomicAdd(count, local); }
  else
  {
    __shared__ unsigned long long sred[4];
    if (lane == 0) sred[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long t = sred[0] + sred[1] + sred[2] + sred[3]; if (t) atomicAdd(&count[(blockIdx.x & 63) * 8], t); }
  }
}

template <int UNROLL>
__global__ __launch_bounds__(256) void v_ballot(const float* __restrict__ map, uint64_t n, float thr, unsigned long long* __restrict__ bits, unsigned long long* count)
{
  const uint64_t n_words = (n + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  unsigned long long local = 0;
  for (uint64_t w = wave * UNROLL; w < n_words; w += n_waves * UNROLL)
  {
    float m[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { const uint64_t idx = (w + u) * 64 + lane; m[u] = idx < n ? map[idx] : -INFINITY; }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      const unsigned long long b = __ballot(m[u] > thr);
      if (w + u < n_words) { if (lane == 0) bits[w + u] = b; local += __popcll(b); }
    }
  }
  if (lane == 0 && local) atomicAdd(count, local);
}

// reference streaming read: float4 sum (what a pure read can reach)
__global__ __launch_bounds__(256) void v_readonly(const float* __restrict__ map, uint64_t n, float thr, unsigned long long* __restrict__ bits, unsigned long long* count)
{
  const float4* map4 = reinterpret_cast<const float4*>(map);
  const uint64_t n4 = n >> 2;
  float acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x)
  { const float4 v = map4[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) bits[0] = 1;
}

typedef void (*kern_t)(const float*, uint64_t, float, unsigned long long*, unsigned long long*);
struct Var { const char* name; kern_t k; int grid; };

int main(int argc, char** argv)
{
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 301752451ull;
  float* map; unsigned long long *bits, *count;
  CHK(hipMalloc(&map, n * 4)); CHK(hipMalloc(&bits, ((n + 63) / 64 + 2) * 8)); CHK(hipMalloc(&count, 8 * 8 * 64));
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (rand() % 100 < 5) ? 0.0f : -740.0f - (rand() % 7);
  for (uint64_t o = 0; o < n; o += h.size()) CHK(hipMemcpy(map + o, h.data(), std::min<uint64_t>(h.size(), n - o) * 4, hipMemcpyHostToDevice));
  std::vector<Var> vars = {
    {"ballot_u8_g2048", v_ballot<8>, 2048}, {"ballot_u8_g8192", v_ballot<8>, 8192},
    {"f4_u4_g2048", v_f4<4, false, true>, 2048}, {"f4_u4_g8192", v_f4<4, false, true>, 8192}, {"f4_u2_g8192", v_f4<2, false, true>, 8192},
    {"f4_u8_g4096", v_f4<8, false, true>, 4096}, {"f4_u4_nt_g8192", v_f4<4, true, true>, 8192}, {"f4_u4_noshuf_g8192", v_f4<4, false, false>, 8192},
    {"f4_u4_spread_g2048", v_f4<4, false, true, true>, 2048}, {"f4_u4_spread_g4096", v_f4<4, false, true, true>, 4096}, {"f4_u8_spread_g2048", v_f4<8, false, true, true>, 2048}, {"f4_u4_spread_g1024", v_f4<4, false, true, true>, 1024}, {"f4_u4_dpp_g1024", v_f4<4, false, true, true, true>, 1024}, {"f4_u4_dpp_g2048", v_f4<4, false, true, true, true>, 2048}, {"f4_u8_dpp_g1024", v_f4<8, false, true, true, true>, 1024}, {"readonly_g2048", v_readonly, 2048}, {"readonly_g16384", v_readonly, 16384},
  };
  {
    unsigned long long *b2; CHK(hipMalloc(&b2, ((n + 63) / 64 + 2) * 8));
    CHK(hipMemset(count, 0, 8 * 8 * 64));
    hipLaunchKernelGGL((v_f4<4, false, true, true, false>), dim3(1024), dim3(256), 0, 0, map, n, -300.0f, bits, count);
    hipLaunchKernelGGL((v_f4<4, false, true, true, true>), dim3(1024), dim3(256), 0, 0, map, n, -300.0f, b2, count);
    CHK(hipDeviceSynchronize());
    const size_t nw = (n + 63) / 64; std::vector<unsigned long long> x(nw), y(nw);
    CHK(hipMemcpy(x.data(), bits, nw * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(y.data(), b2, nw * 8, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < nw; i++) bad += x[i] != y[i];
    printf("dpp vs shuffle words differing: %zu of %zu\n", bad, nw);
    CHK(hipFree(b2));
  }
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  std::vector<std::vector<float>> t(vars.size());
  for (int round = 0; round < 7; round++)
    for (size_t v = 0; v < vars.size(); v++)
    {
      CHK(hipMemset(count, 0, 8 * 8 * 64));
      CHK(hipEventRecord(a));
      hipLaunchKernelGGL(vars[v].k, dim3(vars[v].grid), dim3(256), 0, 0, map, n, -300.0f, bits, count);
      CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
      float ms; CHK(hipEventElapsedTime(&ms, a, b));
      if (round) t[v].push_back(ms);
    }
  for (size_t v = 0; v < vars.size(); v++)
  {
    std::sort(t[v].begin(), t[v].end());
    const double med = t[v][t[v].size() / 2], mn = t[v][0];
    printf("%-22s median %.1f us  min %.1f us  -> %.0f GB/s (median, 4 B/voxel)\n", vars[v].name, med * 1e3, mn * 1e3, n * 4.0 / (med * 1e-3) / 1e9);
  }
  return 0;
}
