// vpf_wave_times.h — measurement hook, compiled ONLY into lab builds (-DVPF_WAVE_TIMES; the product library carries none of it:
// tests/test_abi_cpu.py checks the exported symbols).  Every wave of an instrumented kernel entry leaves one 32-byte record — the constant-
// rate clock (s_memrealtime, 100 MHz: the same on every XCD) at its first and last instruction and at up to three marks a task sets on its
// way (VPF_WAVE_MARK(i)), and the XCC / SE / CU / SIMD it ran on — in a per-TU device buffer, at the slot of its flat wave index in the grid:
// no atomics (a first version drew slots from one counter: 70 000 waves per dispatch serialised on it at 12 ns each and the "measurement" ran
// 4.6 x slower than the kernel — profiles/r05_wave_times_atomic_slots.txt).  A later dispatch overwrites an earlier one: tools/wave_times.py
// runs a train of dispatches and reads the LAST one, which started behind a busy chip like every dispatch of a benchmark loop.
#pragma once
#ifdef VPF_WAVE_TIMES
#include <hip/hip_runtime.h>
namespace vpf {
constexpr uint32_t kWtCap = 1u << 17;
struct WtRec { uint32_t t0_lo, t0_hi_ids, t1_lo, mark[3], waves, spare; };
static __device__ WtRec g_wt[kWtCap];
__device__ inline uint32_t wt_slot() {
  const uint32_t wpb = (blockDim.x * blockDim.y + 63u) >> 6;
  return ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * wpb + (threadIdx.x >> 6)) & (kWtCap - 1u);
}
struct WaveTimer {
  __device__ explicit WaveTimer(uint32_t tag) {
    const uint64_t t0 = wall_clock64();
    if ((threadIdx.x & 63u) == 0) {
      uint32_t hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      WtRec& r = g_wt[wt_slot()];
      r.t0_lo = (uint32_t)t0;
      r.t0_hi_ids = ((uint32_t)(t0 >> 32) & 0xffu) | (xcc & 15u) << 8 | (tag & 15u) << 12 | ((hw >> 4) & 0xfffu) << 16;
      r.mark[0] = r.mark[1] = r.mark[2] = 0u;
      r.waves = gridDim.x * gridDim.y * gridDim.z * ((blockDim.x * blockDim.y + 63u) >> 6);
    }
  }
  __device__ ~WaveTimer() {
    const uint64_t t1 = wall_clock64();
    if ((threadIdx.x & 63u) == 0) g_wt[wt_slot()].t1_lo = (uint32_t)t1;
  }
};
__device__ inline void wt_mark(int i) {  // the first time a wave passes mark i
  const uint64_t t = wall_clock64();
  if ((threadIdx.x & 63u) == 0) { WtRec& r = g_wt[wt_slot()]; if (!r.mark[i]) r.mark[i] = (uint32_t)t | 1u; }
}
static inline uint32_t wave_times_read(void* out, uint32_t cap_records) {  // host: copy the buffer out
  const uint32_t m = cap_records < kWtCap ? cap_records : kWtCap;
  if (out && m && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wt), (size_t)m * sizeof(WtRec)) != hipSuccess) return 0;
  return m;
}
}  // namespace vpf
#define VPF_WAVE_TIMER(tag) const vpf::WaveTimer vpf_wave_timer_(tag)
#ifdef VPF_WAVE_MARKS  // (the marks cost registers and a scalar-memory wait each: the row-band kernel drops from four to three waves per SIMD with them — a second lab build)
#define VPF_WAVE_MARK(i) vpf::wt_mark(i)
#else
#define VPF_WAVE_MARK(i) do { } while (0)
#endif
#define VPF_WAVE_TIMES_EXPORT(name) extern "C" __attribute__((visibility("default"))) uint32_t name(void* out, uint32_t cap) { return vpf::wave_times_read(out, cap); }
#else
#define VPF_WAVE_TIMER(tag) do { } while (0)
#define VPF_WAVE_MARK(i) do { } while (0)
#define VPF_WAVE_TIMES_EXPORT(name)
#endif
