// microbenchmark: latency of a 2 KB producer -> consumer hand-off between two workgroups through global memory, by cache scope and by XCD placement
//   agent scope (what k_program's jobs use: sc1 write-through stores, sc1 loads) against an "L2 scope" that is only correct when producer and consumer share
//   an XCD (plain stores -- the vector L1 writes through to the XCD's L2 -- and sc0 loads, which bypass the consumer's vector L1)
// hipcc --offload-arch=gfx950 -O3 scripts/handoff_probe.hip -o scripts/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE> __device__ __forceinline__ void st64(double *p, double v)
{
  if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
template <int SCOPE> __device__ __forceinline__ double ld64(const double *p)
{
  if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double v;
  if (SCOPE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); // (behind the flag load's buffer_inv)
  return v;
}
template <int SCOPE> __device__ __forceinline__ void sti(int *p, int v)
{
  if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
template <int SCOPE> __device__ __forceinline__ int ldi(const int *p)
{
  if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int v;
  if (SCOPE == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (SCOPE == 2) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); // L1 invalidate, then a plain load
  else asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SCOPE> __global__ __launch_bounds__(256) void k(double *buf, int *flags, int prod, int cons, int iters, unsigned long long *out, int *bad)
{
  const int tid = threadIdx.x;
  if ((int)blockIdx.x != prod && (int)blockIdx.x != cons) return;
  __shared__ int s;
  unsigned long long t0 = 0;
  if ((int)blockIdx.x == prod) {
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; i++) {
      st64<SCOPE>(&buf[tid], (double)(i * 1000 + tid));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) { sti<SCOPE>(&flags[0], i); int n = 0; while (ldi<SCOPE>(&flags[32]) < i && ++n < 200000) __builtin_amdgcn_s_sleep(1); s = n >= 200000; }
      __syncthreads();
      if (s) { if (tid == 0) { atomicAdd(bad, 1000000); sti<0>(&flags[0], 1 << 30); } break; } // every spin is bounded: a hand-off that never arrives ends the run
    }
    if (tid == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t0;
  } else {
    int nbad = 0;
    for (int i = 1; i <= iters; i++) {
      if (tid == 0) { int n = 0; while (ldi<SCOPE>(&flags[0]) < i && ++n < 200000) __builtin_amdgcn_s_sleep(1); s = n >= 200000 || ldi<0>(&flags[0]) >= (1 << 30); }
      __syncthreads();
      if (s) { if (tid == 0) { atomicAdd(bad, 1000000); sti<0>(&flags[32], 1 << 30); } break; }
      const double v = ld64<SCOPE>(&buf[tid]);
      if (v != (double)(i * 1000 + tid)) nbad++;
      __syncthreads();
      if (tid == 0) sti<SCOPE>(&flags[32], i);
    }
    if (nbad) atomicAdd(bad, nbad);
  }
}
int main()
{
  double *buf; int *flags, *bad; unsigned long long *out, h;
  hipMalloc(&buf, 4096); hipMalloc(&flags, 1024); hipMalloc(&out, 8); hipMalloc(&bad, 4);
  const int iters = 2000;
  for (int scope = 0; scope < 4; scope++)
    for (int cons : { 8, 16, 1, 4 }) {
      if (scope >= 1 && cons % 8 != 0 && cons != 1) continue; // one cross-XCD run of the L2 scopes only: they are not coherent there (shown, bounded)
      hipMemset(flags, 0, 1024); hipMemset(bad, 0, 4); hipMemset(buf, 0, 4096);
      if (scope == 0) hipLaunchKernelGGL(k<0>, dim3(32), dim3(256), 0, 0, buf, flags, 0, cons, iters, out, bad);
      else if (scope == 1) hipLaunchKernelGGL(k<1>, dim3(32), dim3(256), 0, 0, buf, flags, 0, cons, iters, out, bad);
      else if (scope == 2) hipLaunchKernelGGL(k<2>, dim3(32), dim3(256), 0, 0, buf, flags, 0, cons, iters, out, bad);
      else hipLaunchKernelGGL(k<3>, dim3(32), dim3(256), 0, 0, buf, flags, 0, cons, iters, out, bad);
      hipDeviceSynchronize();
      int hb; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("%s, producer workgroup 0 -> consumer workgroup %2d (%s): %.2f us per round trip (data 2 KB + flag there, flag back), %d mismatches\n",
             scope == 0 ? "agent scope (sc1)" : scope == 1 ? "L2 scope (plain stores, sc0 loads)" : scope == 2 ? "plain stores, buffer_inv sc0 + plain loads" : "plain stores, buffer_inv sc1 + plain loads", cons, cons % 8 == 0 ? "same XCD" : "other XCD", (double)h * 0.01 / iters, hb);
      fflush(stdout);
    }
  return 0;
}
