// What do LDS operations at random addresses cost a CU?  One workgroup of 1024 threads per CU (the partition and
// count kernels' shape), every thread issuing UNROLL independent operations per iteration on a table of `slots` words:
//   add      ds_add_u32          (result not used)
//   add_rtn  ds_add_rtn_u32      (result used: the counting sort's cursor bump)
//   cas_rtn  ds_cmpst_rtn_b32    (the LDS hash tables' claim)
//   write    ds_write_b32
//   read     ds_read_b32
// Prints cycles per wave-level instruction and CU (clock64 around the loop, one workgroup's view).
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_atomics tools/exp/lds_atomics.hip && build/lds_atomics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kThreads = 1024, kUnroll = 8, kIters = 2000;

template <int OP>
__global__ __launch_bounds__(kThreads) void lds_ops(uint32_t slots_mask, long long *cycles, uint32_t *sink) {
  extern __shared__ uint32_t table[];
  for (uint32_t i = threadIdx.x; i <= slots_mask; i += kThreads) table[i] = OP == 2 ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u, acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < kIters; it++) {
    uint32_t a[kUnroll], r[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; j++) {
      x = x * 1664525u + 1013904223u;
      a[j] = (x >> 9) & slots_mask;
    }
#pragma unroll
    for (int j = 0; j < kUnroll; j++) {
      if (OP == 0) { __hip_atomic_fetch_add(&table[a[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); r[j] = 0; }
      if (OP == 1) r[j] = __hip_atomic_fetch_add(&table[a[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 2) r[j] = atomicCAS(&table[a[j]], 0xFFFFFFFFu, x);
      if (OP == 3) { table[a[j]] = x; r[j] = 0; }
      if (OP == 4) r[j] = table[a[j]];
    }
#pragma unroll
    for (int j = 0; j < kUnroll; j++) acc += r[j];
    if (OP == 3) asm volatile("" ::: "memory");
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  long long *cycles;
  uint32_t *sink;
  hipMalloc(&cycles, sizeof(long long) * cus);
  hipMalloc(&sink, 4);
  const char *names[] = {"add", "add_rtn", "cas_rtn", "write", "read"};
  for (uint32_t slots : {2048u, 32768u}) {
    for (int op = 0; op < 5; op++) {
      const size_t lds = slots * 4;
      auto launch = [&](auto kern) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(kThreads), lds, 0, slots - 1, cycles, sink);
      };
      if (op == 0) launch(lds_ops<0>);
      if (op == 1) launch(lds_ops<1>);
      if (op == 2) launch(lds_ops<2>);
      if (op == 3) launch(lds_ops<3>);
      if (op == 4) launch(lds_ops<4>);
      hipDeviceSynchronize();
      std::vector<long long> h(cus);
      hipMemcpy(h.data(), cycles, sizeof(long long) * cus, hipMemcpyDeviceToHost);
      double mean = 0;
      for (long long c : h) mean += (double)c / cus;
      const double wave_instrs = (double)kIters * kUnroll * (kThreads / 64);
      printf("%6u slots  %-8s  %8.1f cycles per wave instruction and CU  (%.2f lanes per cycle)\n", slots, names[op],
             mean / wave_instrs, 64.0 * wave_instrs / mean);
    }
  }
  return 0;
}
