// k_resize_common.h — what the resize translation units share (k_resize.hip, k_lanczos_mfma.hip): the plane geometry record and the two
// generic batch entries, and the Lanczos-3 tap arithmetic (fp32 weights from fixed fma polynomials, Q14 quantisation) that the oracle
// restates bit for bit (its lanczos_weights_fp32 / lanczos_weights_q14).
#pragma once
#include "k_bilinear_blend.h"
#include "vpf_wave_times.h"

namespace vpf {

// Geometry of one plane of a resize launch, in the form every kernel family takes it.  a0..a3 are family-specific:
//   tiled kernels     a0 = destination rows per tile, a1 = source rows the LDS layout is sized for, a2 = 16-B units per staged source row,
//                     a3 = log2(lanes per row while staging)
//   row-pair kernels  a0 = strip size in 16-B units
//   half3_r16         a0 = 1024-px chunks per row, a1 = tasks
struct PlaneGeom {
  uint32_t sw, sh, dw, dh;
  float scx, scy;
  int vec_ok;
  uint32_t a0, a1, a2, a3;
};
// A resize "Task" is a struct with `static constexpr int kThreads` and
//   static VPF_DEV void run(const uint8_t* src, uint32_t sp, uint8_t* dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by)
// The single-frame kernels below keep their scalar-argument entries (kernarg preload, see VPF_ONE_SRC_PARAMS in vpf_internal.h);
// vpf_resize_batch reaches the same task bodies through these two generic entries:
//   k_plane_batch   ONE plane (index k in FrameDesc) of up to 32 frames: blockIdx.z = frame
//   k_planes_mp     EVERY plane of up to 32 frames in one dispatch: blockIdx.z = frame, blockIdx.y runs through the planes' block rows
//                   one plane after the other (by0[p] = first blockIdx.y of plane p), blockIdx.x covers the widest plane (a task returns
//                   at once when its block lies outside its plane)
// Workgroups are handed to the chip's eight XCDs round robin in launch order, and every XCD has an L2 of its own.  With the plain numbering
// the blocks next to each other in a picture — which share the 128-B lines their strips straddle, and, above each other, the source rows
// their bands / taps overlap on — sit on eight different L2s, and every shared line comes from HBM once per sharer (measured on the
// matrix-core Lanczos kernel: HBM reads 1.30-1.33 x the source -> 1.02-1.09 x with the renumbering; profiles/r03_pmc_lanczos_traffic_*).
// picture_order() renumbers: XCD x takes the x-th eighth of the picture-ordered block list (x fastest, then y, then the frame), so neighbours
// in the picture are neighbours in time on ONE L2.  A bijection of the launch grid onto itself for any grid size.
struct BlockId {
  uint32_t x, y, z;
};
VPF_DEV BlockId picture_order() {
  const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), xcd = lin & 7u, idx = lin >> 3, per = total >> 3, rem = total & 7u;
  const uint32_t m = xcd < rem ? xcd * (per + 1u) + idx : rem * (per + 1u) + (xcd - rem) * per + idx;
  const uint32_t yz = m / gx;
  return BlockId{m - yz * gx, yz % gy, yz / gy};
}
template <class Task, class BA = BatchArgs>  // BA: the frame table's size (BatchArgs: <= 32 frames, BatchArgsL: <= 128; vpf_internal.h)
__global__ __launch_bounds__(Task::kThreads) void k_plane_batch(const BA args, const int k, const PlaneGeom G) {
  VPF_WAVE_TIMER(1);
  const BlockId b = picture_order();
  const FrameDesc& f = args.f[b.z];
  Task::run(f.s[k], f.sp[k], f.d[k], f.dp[k], G, b.x, b.y);
}
struct PlaneTable {
  PlaneGeom g[3];
  uint32_t by0[3], k[3], ch[3], np;
};
#ifdef VPF_WAVE_TIMES  // (lab builds: the timer's common exit costs the band kernels 30 VGPRs and with them the fourth wave per SIMD — held at the product's occupancy)
#define VPF_WT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define VPF_WT_OCCUPANCY
#endif
template <template <int> class TaskCH, class BA = BatchArgs>
__global__ __launch_bounds__(TaskCH<3>::kThreads) VPF_WT_OCCUPANCY void k_planes_mp(const BA args, const PlaneTable T) {
  VPF_WAVE_TIMER(2);
  const BlockId b = picture_order();
  const FrameDesc& f = args.f[b.z];
  const uint32_t by = b.y;
  const uint32_t pi = (uint32_t)(T.np > 1 && by >= T.by0[1]) + (uint32_t)(T.np > 2 && by >= T.by0[2]);
  const uint32_t k = T.k[pi], lby = by - T.by0[pi];
  switch (T.ch[pi]) {  // workgroup-uniform
    case 1: TaskCH<1>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], b.x, lby); break;
    case 2: TaskCH<2>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], b.x, lby); break;
    default: TaskCH<3>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], b.x, lby); break;
  }
}

#ifdef VPF_LAB_FORMS  // measured and not selected by any policy: built into tools/lab/libvpfhip_forms.so only
// ------------------------------------------------------------------------------------------
// The persistent launch of the band kernels, second form (round 6).  Round 5's drew ONE ticket per 4-row wave item, waited for
// it, ran the item, and asked again (17 280 serialised round trips for a 32-frame dispatch of 720p luma planes, plus eight failing fetches per
// wave at the end): ten times slower than the grid.  This one is what VERDICT r4 / r5 asked for:
//   * the unit of work is a CHUNK — `chunk` consecutive bands of one column chunk of one plane of one frame — and a wave walks down it like the
//     march form does (the next band's source rows requested before this band is blended);
//   * a wave's FIRST chunk is assigned statically (workgroup b, wave w -> chunk (b >> 3) * waves-per-group + w of XCD b & 7's share): no wave
//     starts by waiting for a counter;
//   * the ticket for the chunk AFTER the next is drawn (one lane, device-scope atomicAdd) when a chunk starts, and read a chunk later, when the
//     wave has long been waiting on its own source rows anyway: the counter's latency is hidden;
//   * the NEXT chunk's first source rows are requested before this chunk's last band is blended and stored: a wave runs through chunks the way
//     the march form runs through bands, whatever frame, plane or column the next chunk belongs to (the column taps are rebuilt after the blend);
//   * one counter per XCD as before (XCD x works through the x-th eighth of the picture-ordered chunk list: neighbours in the picture are
//     neighbours in time on ONE L2); a wave whose own counter has run dry visits `hops - 1` neighbours', looking (plain load) before it draws.
// Counters: two sets of eight per slot (vpf_persist.h).  A launch draws from one — zero when it starts — and puts the OTHER back to zero for
// the stream's next persistent launch, which cannot start before this one has ended: no last-ticket protocol, failing fetches are harmless.
// A task takes part through
//   static VPF_DEV void run_chunks(BandChunk& c, Stream& ts, const PlaneTable& T)  — works through c and whatever ts.next() hands out while
//   it has this task's channel count; leaves the first chunk of another channel count (or nb = 0: none left) in c
// ------------------------------------------------------------------------------------------
struct PersistArgs {
  uint32_t* ctr;        // this launch's eight counters (zero when it starts)
  uint32_t* ctr_other;  // the slot's other eight: zeroed here for the stream's next persistent launch
  uint32_t lo[9];       // XCD x owns chunks [lo[x], lo[x + 1])
  uint32_t per_frame;   // chunks per frame (all planes)
  uint32_t p0[3];       // first chunk of plane p inside a frame
  uint32_t nbx[3];      // column chunks of plane p
  uint32_t nbands[3];   // bands of plane p
  uint32_t chunk;       // bands per chunk
  uint32_t hops;        // counters a wave visits (1: its own XCD's only)
};
struct BandChunk {  // wave-uniform
  const uint8_t* src;
  uint8_t* dst;
  uint32_t sp, dp, pi, bx, band0, nb;  // plane index, column chunk, first band, bands (0: no more work)
};
template <class BA, int WPG /* waves per workgroup */>
struct ChunkStream {
  const BA& args;
  const PlaneTable& T;
  const PersistArgs& P;
  uint32_t xcc, hop, x, lo, left;  // the counter being drawn from: XCD x's share starts at chunk `lo`; `left` of it lie behind the statically assigned ones
  uint32_t tn;                     // lane 0: the ticket in flight
  uint32_t sidx;                   // hops == 0 (measurement: no counters at all): the wave's chunks are sidx, sidx + waves of its XCD, ...
  float rpf;
  VPF_DEV ChunkStream(const BA& a, const PlaneTable& t, const PersistArgs& p) : args(a), T(t), P(p), hop(0), tn(0) {
    xcc = blockIdx.x & 7u;  // workgroups go to the XCDs round robin in launch order (picture_order() relies on the same); a share is drained by the
                            // workgroups that NAME it whichever XCD they really run on — completeness does not depend on the placement
    rpf = 1.0f / (float)P.per_frame;
  }
  // chunks of XCD x's share that are handed out statically: one per wave of the workgroups b with b & 7 == x
  VPF_DEV uint32_t statics(uint32_t x_) const {
    const uint32_t nx = P.lo[x_ + 1] - P.lo[x_], w = gridDim.x > x_ ? ((gridDim.x - x_ + 7u) >> 3) * (uint32_t)WPG : 0u;
    return w < nx ? w : nx;
  }
  VPF_DEV void open(uint32_t x_) {
    x = x_;
    const uint32_t st = statics(x_);
    lo = P.lo[x_] + st;
    left = P.lo[x_ + 1] - lo;
  }
  VPF_DEV void draw() {
    if ((threadIdx.x & 63u) == 0) tn = __hip_atomic_fetch_add(P.ctr + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  VPF_DEV BandChunk decode(uint32_t it) const {
    // chunk -> (frame, plane, chunk row, column chunk); (it + 0.5) / per_frame is >= 0.5 / per_frame away from an integer: exact in fp32 for it < 2^22
    uint32_t z = (uint32_t)(((float)it + 0.5f) * rpf);
    uint32_t r = it - z * P.per_frame;
    if ((int32_t)r < 0) { z--; r += P.per_frame; } else if (r >= P.per_frame) { z++; r -= P.per_frame; }
    z = __builtin_amdgcn_readfirstlane(z); r = __builtin_amdgcn_readfirstlane(r);
    const uint32_t pi = (uint32_t)(T.np > 1 && r >= P.p0[1]) + (uint32_t)(T.np > 2 && r >= P.p0[2]);
    r -= P.p0[pi];
    const uint32_t nbx = P.nbx[pi], crow = r / nbx, k = T.k[pi], band0 = crow * P.chunk, nbands = P.nbands[pi];
    const FrameDesc& f = args.f[z];
    return BandChunk{f.s[k], f.d[k], f.sp[k], f.dp[k], pi, r - crow * nbx, band0, nbands - band0 < P.chunk ? nbands - band0 : P.chunk};
  }
  VPF_DEV BandChunk first() {
    const uint32_t idx = (blockIdx.x >> 3) * (uint32_t)WPG + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    open(xcc);
    sidx = idx;
    if (P.hops) draw();  // (read by the first next(): a chunk from now)
    if (idx < P.lo[xcc + 1] - P.lo[xcc]) return decode(P.lo[xcc] + idx);
    return next();
  }
  VPF_DEV BandChunk next() {
    if (!P.hops) {  // static stride
      sidx += ((gridDim.x - xcc + 7u) >> 3) * (uint32_t)WPG;
      if (sidx < P.lo[xcc + 1] - P.lo[xcc]) return decode(P.lo[xcc] + sidx);
      return BandChunk{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    }
    for (;;) {
      if (hop >= P.hops) return BandChunk{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
      const uint32_t t = __builtin_amdgcn_readfirstlane(tn);
      if (t < left) {
        draw();
        return decode(lo + t);
      }
      for (;;) {  // this counter has run dry: the next one that still has chunks (a look first: a dry counter costs a load, not an atomic)
        if (++hop >= P.hops) return BandChunk{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
        open((xcc + hop) & 7u);
        uint32_t seen = 0;
        if ((threadIdx.x & 63u) == 0) seen = __hip_atomic_load(P.ctr + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(seen) < left) break;
      }
      draw();
    }
  }
};
template <template <int> class TaskCH, class BA = BatchArgs>
__global__ __launch_bounds__(TaskCH<3>::kThreads) void k_planes_mp_persist(const BA args, const PlaneTable T, const PersistArgs P) {
  VPF_WAVE_TIMER(6);
  if (blockIdx.x == 0 && threadIdx.x < 8) __hip_atomic_store(P.ctr_other + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ChunkStream<BA, TaskCH<3>::kThreads / 64> ts(args, T, P);
  BandChunk c = ts.first();
  while (c.nb) {
    switch (T.ch[c.pi]) {  // wave-uniform
      case 1: TaskCH<1>::run_chunks(c, ts, T); break;
      case 2: TaskCH<2>::run_chunks(c, ts, T); break;
      default: TaskCH<3>::run_chunks(c, ts, T); break;
    }
    wave_lds_sync();  // the chunk's LDS reads are done before the next task's rows overwrite the strips
  }
}
#endif  // VPF_LAB_FORMS

// four Q12 values (|w| < 2^27) -> four bytes: clamp(w >> 12, 0, 255), value k in byte k.  Six instructions instead of nine (four shifts, two
// v_cvt_pk_i16_i32, two v_sat_pk_u8_i16, one v_perm_b32): the SDWA forms write a 16-bit result into the upper half of a register whose
// lower half already holds its neighbour, so nothing has to be packed afterwards (w >> 12 fits 16 bits as it is).  The inputs are results
// of ordinary VALU instructions of the compiler's, not of an MFMA: nothing in here needs wait states the compiler cannot see, except the
// gfx940+ rule that a VALU reading a register right after an SDWA wrote part of it waits one state (the s_nop / the instruction order).
VPF_DEV uint32_t shift12_sat_pack4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  uint32_t a, b, r;
  asm("v_ashrrev_i32_e32 %0, 12, %3\n\t"
      "v_ashrrev_i32_e32 %1, 12, %5\n\t"
      "v_ashrrev_i32_sdwa %0, 12, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_ashrrev_i32_sdwa %1, 12, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_e32 %2, %0\n\t"
      "v_sat_pk_u8_i16_sdwa %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(a), "=&v"(b), "=&v"(r)
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  return r;
}

// the same six instructions in two halves, so that a caller can put an MFMA between them (k_lanczos_mfma.hip interleaves its pack with the
// next tile's MFMAs): a = {w0 >> 12, w1 >> 12}, b.lo = w2 >> 12 | then b.hi = w3 >> 12, r = sat(a) | sat(b) << 16
VPF_DEV void shift12_sat_pack4_a(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t& a, uint32_t& b) {
  asm("v_ashrrev_i32_e32 %0, 12, %2\n\t"
      "v_ashrrev_i32_e32 %1, 12, %4\n\t"
      "v_ashrrev_i32_sdwa %0, 12, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(a), "=&v"(b)
      : "v"(w0), "v"(w1), "v"(w2));
}
VPF_DEV uint32_t shift12_sat_pack4_b(uint32_t a, uint32_t b, uint32_t w3) {
  uint32_t r;
  asm("v_ashrrev_i32_sdwa %1, 12, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_e32 %0, %2\n\t"
      "v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(r), "+v"(b)
      : "v"(a), "v"(w3));
  return r;
}

// ------------------------------------------------------------------------------------------
// Lanczos-3 taps (see the comment above LanczosGatherTask in k_resize.hip)
// ------------------------------------------------------------------------------------------
VPF_DEV float lz_sinpi_poly(float g) {  // sin(pi g), g in [0, 0.5]
  const float x = 3.14159274f * g, x2 = x * x;
  float p = __builtin_fmaf(x2, -2.50521084e-8f, 2.75573192e-6f);
  p = __builtin_fmaf(x2, p, -1.98412698e-4f);
  p = __builtin_fmaf(x2, p, 8.33333333e-3f);
  p = __builtin_fmaf(x2, p, -1.66666667e-1f);
  p = __builtin_fmaf(x2, p, 1.0f);
  return x * p;
}
VPF_DEV float lz_cos_poly(float x) {  // cos(x), x in [0, pi/3]
  const float x2 = x * x;
  float p = __builtin_fmaf(x2, -2.75573192e-7f, 2.48015873e-5f);
  p = __builtin_fmaf(x2, p, -1.38888889e-3f);
  p = __builtin_fmaf(x2, p, 4.16666667e-2f);
  p = __builtin_fmaf(x2, p, -0.5f);
  return __builtin_fmaf(x2, p, 1.0f);
}
struct LTap {
  int32_t i0;
  float w[6];
};
VPF_DEV LTap make_ltap(uint32_t d, float scale) {
  LTap t;
  const float s = __builtin_fmaf((float)d + 0.5f, scale, -0.5f);
  const float fl = __builtin_floorf(s);
  t.i0 = (int32_t)fl;
  const float f = s - fl;
  if (f == 0.f) {
    t.w[0] = t.w[1] = t.w[3] = t.w[4] = t.w[5] = 0.f; t.w[2] = 1.f;
    return t;
  }
  const float s1 = lz_sinpi_poly(f <= 0.5f ? f : 1.0f - f);
  const float s3 = lz_sinpi_poly(f * 0.333333343f);
  const float c3 = lz_cos_poly(1.04719758f * f);
  constexpr float cm[6] = {-0.5f, 0.5f, 1.0f, 0.5f, -0.5f, -1.0f};
  constexpr float sm[6] = {-0.866025388f, -0.866025388f, 0.0f, 0.866025388f, 0.866025388f, 0.0f};
  constexpr float sg[6] = {1.0f, -1.0f, 1.0f, -1.0f, 1.0f, -1.0f};
  // w_k = L(t_k) / sum_j L(t_j) with L(t) = 3 sin(pi t) sin(pi t / 3) / (pi t)^2, t_k = f - (k - 2): multiplying numerator and
  // denominator by prod_j t_j^2 leaves n_k D_k / sum_j n_j D_j with n_k = sin(pi t_k) sin(pi t_k / 3) and D_k = prod_{j != k} t_j^2
  // — ONE division per weight set instead of seven (the weights are ~1/4 of the tiled kernel's instructions at small tiles)
  float n[6], u[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float tt = f - (float)(k - 2);
    u[k] = tt * tt;
    n[k] = (sg[k] * s1) * __builtin_fmaf(s3, cm[k], -(c3 * sm[k]));
  }
  float pre[6], suf[6];  // pre[k] = u_0 .. u_{k-1}, suf[k] = u_{k+1} .. u_5
  pre[0] = 1.0f; suf[5] = 1.0f;
#pragma unroll
  for (int k = 1; k < 6; k++) pre[k] = pre[k - 1] * u[k - 1];
#pragma unroll
  for (int k = 4; k >= 0; k--) suf[k] = suf[k + 1] * u[k + 1];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    t.w[k] = n[k] * (pre[k] * suf[k]);
    sum += t.w[k];
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < 6; k++) t.w[k] *= inv;
  return t;
}

// The horizontal pass on 8-bit surfaces runs in integers: the six normalised weights become Q14 fixed point (ties to even), tap 2
// absorbs the rounding residue so that they sum to exactly 16384 (a flat picture stays flat), and H = sum q_k p_k is exact in 32
// bits whatever the order — which is what lets the tiled kernel take two taps per v_dot2_i32_i16 and still match the gather form
// and the oracle bit for bit.  Every |q_k| <= 16384 fits an int16.
struct QTap {
  int32_t i0;
  int32_t q[6];
};
VPF_DEV QTap quantize_ltap(const LTap& t) {
  QTap o;
  o.i0 = t.i0;
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) { o.q[k] = (int32_t)__builtin_rintf(t.w[k] * 16384.0f); sum += o.q[k]; }
  o.q[2] += 16384 - sum;
  return o;
}
VPF_DEV int32_t ltap_i0(uint32_t d, float scale) {  // make_ltap's first expression sequence
  return (int32_t)__builtin_floorf(__builtin_fmaf((float)d + 0.5f, scale, -0.5f));
}

}  // namespace vpf
