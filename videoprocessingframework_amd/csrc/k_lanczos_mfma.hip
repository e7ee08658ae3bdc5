// k_lanczos_mfma.hip — Lanczos-3 resize of 8-bit planes on the integer matrix cores (gfx950: v_mfma_i32_16x16x64_i8).
//
// Replaces nppiResize_8u_C3R / _C1R with NPPI_INTER_LANCZOS — the filter the reference's resizer asks for
// (NppResizeSurfacePacked3C_Impl::Run / NppResizeSurfacePlanar_Impl::Run, src/TC/src/Tasks.cpp:1162-1203,1217-1261).
//
// A separable resize is two banded matrix products: H = S Wx (every source row against the column weights) and O = Wy H.  With six
// taps per output sample the bands are narrow, and on the vector ALU the filter is issue-bound long before it is memory-bound (round 2:
// 0.22-0.50 of the HBM roofline, half of it v_perm_b32 / v_dot2 tap extraction).  The matrix cores do not care that most of a band is
// zero — an i8 MFMA retires 16 x 16 x 64 products in 16 cycles — and, decisive here, they take the SOURCE BYTES AS THEY LIE IN MEMORY as
// an operand: no unpacking, no per-channel shuffles.  The whole filter is integer arithmetic (the test oracle restates it: resize_plane_lanczos,
// FP32 mode), so the order in which an MFMA adds its 64 products does not matter and every kernel of the family writes the same bytes:
//   H  = sum_k qx[k] s[k]                          Q14 weights (sum 16384), exact
//   Hr = (H + 128) >> 8                            Q6, fits 16 bits with the Lanczos overshoot
//   z  = Hr - 8192 = 256 zh + zl,  q = 256 qh + ql    signed low bytes; vertical taps that clamp onto the same source row are merged first
//   V  = 2^19 + sum_k (qy[k] z[k] - ql[k] zl[k]) >> 8      i.e. 256 sum qh zh + sum (ql zh + qh zl): the lowest partial product is not formed
//   out = clamp((V + 2^11) >> 12, 0, 255)
//
// Work decomposition (tests/lanczos_mfma_model.py is an executable model of exactly this bookkeeping, checked against the oracle on the CPU):
//   * a WAVE owns a strip of NT "N-tiles" of 16 destination BYTES (byte b = pixel b / CH, channel b % CH: packed RGB needs no special
//     case) and a band of destination rows, and marches down the source in tiles of 16 rows.  The four waves of a workgroup own four
//     neighbouring strips of the same band and share the row-weight operands (one barrier per 64 destination rows).
//   * pass 1, per source tile T and N-tile j:  D[16 rows][16 bytes] = A B with A = the 64 source bytes of each row that start at the
//     tile's window ws_j (one ds_read_b128 per lane: lane (i, g) reads row i, bytes 16 g .. 16 g + 15; bytes are staged with 0x80 xor-ed
//     in = s - 128 as a signed byte) and B = the column weights, each Q14 weight as two signed bytes (w = 256 wh + wl -> two MFMAs HI, LO).
//     Clamped taps simply add their weights on the edge pixel's slot, so image edges cost nothing.  Per lane:
//     h'' = ((HI + 128) << 8) + LO + 128 (the 128s ride in as the MFMAs' C operand) holds z + 128 in bits 8 .. 23, and one v_perm_b32 + one
//     xor per ROW PAIR pack the four rows a lane holds into two dwords of (zl, zh) byte pairs.
//   * the D layout of pass 1 (lane (n, g) holds rows 4 g .. 4 g + 3 of column n) IS the A layout of pass 2 (lane (n, g) holds K slots
//     16 g .. 16 g + 15 of row n) once the K slots are numbered to match: the tile in ring slot p = (T - t_first) & 3 sits in K chunk p >> 1,
//     and its source row 4 g + r owns the byte pair 16 g + 8 (p & 1) + 2 r (zl) / + 1 (zh).  The ring (the byte pairs of the last four
//     source tiles, 2 x 4 VGPRs per N-tile) never moves: the march is unrolled four deep and pass 1 is instantiated per ring slot.
//   * pass 2, per destination tile of 16 rows and N-tile: D[16 bytes][16 rows] = A B with A = the ring (two K chunks) and B = the row
//     weights: the Y operand carries qh against the zl slots and ql against the zh slots, the X operand — qh against zh — is derived from
//     it ((Y << 8) & 0xff00ff00); out = ((X << 8) + Y) >> 12 with 2^19 + 2^11 riding in as Y's C operand.  A lane ends up with four
//     horizontally adjacent output bytes -> one dword (shift12_sat_pack4: six instructions) -> a wave-private LDS tile -> dense 16-B row
//     stores.  In both passes the MFMAs of N-tile j + 1 are issued ahead of the vector-ALU work on tile j.
//   * weights: operand images depend on the plane shape only; two small kernels build them once per shape into a table in static device
//     memory (see "Weight operands" below), the main kernel loads its strip's 2 NT column operands (16 B per lane each) and copies the row
//     operands of 64 destination rows to LDS by LDS-DMA.  Without a table (arena full / tuning flag) the same code evaluates them in place.
//   * blocks are numbered XCD-aware (picture_order, k_resize_common.h): neighbouring strips and bands run on one L2.
//   * two-chunk windows (KC = 2, 4-tile strips): where the taps of a tile's 16 bytes spread over more than 64 source bytes — horizontal
//     factors of ~2.2 .. 6, the resize in front of a network — the window is 128 bytes and every pass-1 product two chained MFMAs (K chunk 1
//     accumulates onto chunk 0); column operands [hi | lo][K chunk][tile].  Everything after pass 1 is the same.
//   * half tiles (G.a3 = 3): where 16 destination rows would need more than the ring's four source tiles — vertical factors of ~2.9 .. 6 —
//     a tile carries 8 destination rows (rows 8 .. 15 of the MFMA tile repeat row 7 and are never stored): lzm_group_row.
// VGPRs: 2 x 4 x NT x KC column operands + 2 x 4 x NT ring + two staging sets: NT = 8 -> two waves per SIMD, NT = 4 -> three (KC = 2: two).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <vector>
#include <type_traits>

#include "k_resize_common.h"
#include "vpf_lzm_plan.h"
#include "vpf_plan_bounds.h"

namespace vpf {

typedef int v4i __attribute__((ext_vector_type(4)));

// taps i0 - 2 .. i0 + 3 clamped to [0, size - 1]; the weights of taps that fall on the same sample are summed into the LAST of them and
// the others marked dead (pos = -1): every live tap of a set has its own source sample
struct MTap {
  int32_t pos[6];
  int32_t q[6];
};
VPF_DEV MTap merge_taps(const QTap& t, uint32_t size) {
  MTap m;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int32_t i = t.i0 + k - 2;
    m.pos[k] = i < 0 ? 0 : (i > (int32_t)size - 1 ? (int32_t)size - 1 : i);
    m.q[k] = t.q[k];
  }
#pragma unroll
  for (int k = 0; k < 5; k++)
    if (m.pos[k] == m.pos[k + 1]) { m.q[k + 1] += m.q[k]; m.pos[k] = -1; }
  return m;
}
// w = 256 hi + lo, lo in [-128, 127] (|w| < 32512: merged Lanczos weights stay below 1.3 x 16384)
VPF_DEV void split_i8(int32_t w, int32_t& hi, int32_t& lo) {
  lo = ((w + 128) & 0xff) - 128;
  hi = (w - lo) >> 8;
}
// keeps the compiler from re-associating a chain of (a << n) + b into shift, shift, add3: an EMPTY asm (no instruction, so none of the
// MFMA -> VALU wait states the compiler pads for its own instructions — and not for the contents of an asm — can go missing)
VPF_DEV uint32_t opaque(uint32_t v) {
  asm("" : "+v"(v));
  return v;
}
// PF = staging loads a lane keeps in flight = ceil(16-B units per staged row / 4); KC = 64-B K chunks of a pass-1 window (2: the taps of a
// tile's 16 bytes spread over up to 128 source bytes — horizontal factors of ~2.2 .. 6 — and pass 1 chains two MFMAs per product)
// UP2 = the ring of TWO, for up-scales (every destination tile finds its source rows in the source tiles (Tmax - 1, Tmax):
// vpf_bound_lzm_rows_two): pass 2 is ONE K chunk — two MFMAs per N-tile instead of four — and the register file holds ONE chunk per N-tile,
// (previous tile, this tile): pass 1 moves the second half down (two v_mov per N-tile and source tile) and writes the new tile's (zl, zh)
// pairs into the second; a destination tile multiplies it by a one-chunk weight operand (its own row-table layout: lzm_row_group).  A 2x
// up-scale emits two destination tiles per source tile: 10 MFMAs per N-tile and source tile become 6 — the matrix pipe stops being what
// the waves of a SIMD queue for — and 32 VGPRs of ring are gone: three workgroups per CU instead of two
template <int CH, int NT, int PF, int KC = 1, bool UP2 = false>
struct LanczosMfmaTask {
  static constexpr int kThreads = 256;
  // the register diet (one tile in flight, one A-operand address, two A operands ahead): with the ring of two it is worth a workgroup per CU (the
  // 8-tile form: 166 VGPRs, three; the 4-tile form: 116, four).  Tried on the ring-of-four 4-tile strips too: 164 -> 145, still three — not taken
  static constexpr bool kLean = UP2 && PF == 2;  // (the wide ring-of-two strips — 8 tiles at 1.5 x, rows of up to 256 B — keep the full prefetch at two workgroups per CU)
  static constexpr int kGroupsPerCu = (NT == 4 && KC == 1) || (UP2 && PF == 2) ? 3 : 2;  // register budget: 168 / 256 VGPRs (the 4-tile ring of two fits four: its LDS decides)
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by,
                          const u32x4* __restrict__ ctab, const u32x4* __restrict__ rtab);  // the shape's column / row weight tables (nullptr: evaluate in place)
};

// ------------------------------------------------------------------------------------------------------------------------------------
// Weight operands.  They depend on the plane SHAPE only — not on the frame, and the column operands not on the band either — yet a
// 32-frame launch of 1080p -> 720p evaluated each column set 64 times and each row set 256 times: 22 % of the kernel
// (profiles/r03_lanczos_ablation.txt, "setup only").  They are built ONCE per shape by two small kernels into a table, and the main kernel
// loads them — 16 B per lane and operand — instead of evaluating ~130 instructions per set.  Where the tables live (vpf_lzm_plan.h):
// in the caller's workspace (vpf_resize_ws: NPP's scratch-buffer pattern, what the Task layer's ResizeSurface uses), else in a small
// static arena of this library (4 MiB per device, least-recently-used eviction guarded by events) — the C ABI still allocates nothing.
// Without a table (a shape larger than the arena, VPF_TUNE_RESIZE_MFMA | 0x10000) the same code evaluates the weights in place: same
// pixels.  Builds compose an image in LDS and copy it out whole: concurrent builds of a shape only ever write the final bytes.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kLzmArenaBytes = 4u << 20;
__device__ u32x4 g_lzm_arena[kLzmArenaBytes / 16];

// window of the N-tile that starts at destination byte b: the 16-B aligned source byte below the first tap of its first pixel (tiles
// past the row end copy the last window)
template <int CH>
VPF_DEV uint32_t lzm_window(uint32_t b, uint32_t dwb, uint32_t sw, float scx) {
  const uint32_t bb = b < dwb - 1 ? b : dwb - 1;
  int32_t p0 = ltap_i0(bb / CH, scx) - 2;
  p0 = p0 < 0 ? 0 : (p0 > (int32_t)sw - 1 ? (int32_t)sw - 1 : p0);
  return ((uint32_t)CH * (uint32_t)p0) & ~15u;
}

// column weights of the strip that starts at destination byte ob0 -> pass-1 B operands (b1h / b1l [K chunk * NT + tile]), kLzmB1Chunk / KC
// tiles at a time through an 8-KiB LDS scratch.  Operand image in LDS: [hi | lo][K chunk][tile][lane 16 g + n][16 bytes]; zero it, evaluate
// the sets of the chunk's pixels (one per lane), scatter their bytes, read the operands back.  Clamped taps add their weights on the edge
// pixel's slot: image edges cost nothing.
template <int CH, int NT, int KC = 1>
VPF_DEV void lzm_col_operands(uint8_t* lds, uint32_t lane, uint32_t ob0, uint32_t dwb, uint32_t sw, float scx, v4i (&b1h)[NT * KC], v4i (&b1l)[NT * KC]) {
  constexpr int TPC = (int)kLzmB1Chunk / KC > NT ? NT : ((int)kLzmB1Chunk / KC ? (int)kLzmB1Chunk / KC : 1);  // tiles per pass through the scratch (4, 2, 1 for KC = 1, 2, 3)
  constexpr uint32_t PLANE = KC * TPC * 1024u;     // bytes of the hi image (<= kLzmB1Chunk KiB)
  static_assert(NT % TPC == 0 && 2u * PLANE <= 2u * kLzmB1Chunk * 1024u, "the scratch holds both images of a pass");
  const uint32_t ob1 = ob0 + 16u * NT < dwb ? ob0 + 16u * NT : dwb;
#pragma unroll
  for (int ck = 0; ck < NT / TPC; ck++) {
    const uint32_t cb0 = ob0 + 16u * TPC * ck, cb1 = cb0 + 16u * TPC < ob1 ? cb0 + 16u * TPC : ob1;  // destination bytes of the chunk
    u32x4* z = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < (int)(2u * PLANE / 1024u); i++) z[i * 64 + lane] = u32x4{0, 0, 0, 0};
    if (cb0 < cb1) {
      const uint32_t px_first = cb0 / CH, px_last = (cb1 - 1) / CH;
      for (uint32_t px = px_first + lane; px <= px_last; px += 64) {
        const MTap m = merge_taps(quantize_ltap(make_ltap(px, scx)), sw);
        int32_t whi[6], wlo[6];
#pragma unroll
        for (int k = 0; k < 6; k++) split_i8(m.q[k], whi[k], wlo[k]);
#pragma unroll
        for (int c = 0; c < CH; c++) {
          const uint32_t b = px * CH + c;
          if (b < cb0 || b >= cb1) continue;
          const uint32_t j = (b - cb0) >> 4, n = (b - cb0) & 15;
          const uint32_t wsj = lzm_window<CH>(cb0 + 16u * j, dwb, sw, scx);
          uint8_t* const cell = lds + (j * 64 + n) * 16;
#pragma unroll
          for (int k = 0; k < 6; k++) {
            if (m.pos[k] < 0) continue;
            const uint32_t kk = (uint32_t)CH * (uint32_t)m.pos[k] + c - wsj;  // < 64 KC (vpf_bound_lzm_span_win)
            uint8_t* const a = cell + (kk >> 6) * (TPC * 1024u) + ((kk >> 4) & 3u) * 256 + (kk & 15);
            a[0] = (uint8_t)whi[k];
            a[PLANE] = (uint8_t)wlo[k];
          }
        }
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int kc = 0; kc < KC; kc++)
#pragma unroll
      for (int j = 0; j < TPC; j++) {
        b1h[kc * NT + ck * TPC + j] = *reinterpret_cast<const v4i*>(lds + kc * (TPC * 1024u) + (j * 64 + lane) * 16);
        b1l[kc * NT + ck * TPC + j] = *reinterpret_cast<const v4i*>(lds + PLANE + kc * (TPC * 1024u) + (j * 64 + lane) * 16);
      }
    wave_lds_sync();
  }
}

VPF_DEV int32_t lzm_band_first_tile(uint32_t ya, float scy, uint32_t sh) {  // first source tile of the band that starts at destination row ya
  int32_t r = ltap_i0(ya, scy) - 2;
  r = r < 0 ? 0 : (r > (int32_t)sh - 1 ? (int32_t)sh - 1 : r);
  return r >> 4;
}

// row weights of group g (the 64 destination rows from ya + 64 g; rows past the band repeat its last row and are never stored) of the
// band [ya, yb] -> the pass-2 operand image wm (kLzmWmBytes).  Lane = row; the bytes are SCATTERED into the image: destination tile
// t = lane >> 4 owns [chunk c][lane 16 g' + y][16 B] of the Y operand; the source tile in ring slot p = (T - t_first) & 3 sits in chunk p >> 1,
// and source row 16 T + 4 g' + r owns the byte pair 8 (p & 1) + 2 r (zl) / + 1 (zh): Y carries qh against zl and ql against zh (the X operand —
// qh against zh, nothing against zl — is (Y << 8) & 0xff00ff00: pass 2 derives it)
// rts = log2 of the destination rows a 16-row MFMA tile really carries: 4, or 3 — HALF tiles, for vertical factors of ~2.9 .. 6 where 16 rows
// would need more than the ring's four source tiles: rows 8 .. 15 of every tile then repeat its row 7 and are never stored
VPF_DEV uint32_t lzm_group_row(uint32_t lane, uint32_t ya, uint32_t yb, uint32_t g, uint32_t rts) {  // the destination row of lane (tile lane >> 4, row lane & 15) of group g
  const uint32_t rt = 1u << rts, rr = (lane & 15u) < rt ? (lane & 15u) : rt - 1u, row = ya + (g << (rts + 2u)) + ((lane >> 4) << rts) + rr;
  return row < yb ? row : yb;
}
// up2 (the ring of two, rts == 4 only): ONE chunk per destination tile — the tile's last source tile Tmax (what the march's emit test reads:
// the last tap of the tile's last row) sits in the chunk's second half, Tmax - 1 in its first; no tap lies elsewhere (host: vpf_bound_lzm_rows_two)
VPF_DEV void lzm_row_group(uint8_t* wm, uint32_t lane, uint32_t ya, uint32_t yb, uint32_t g, float scy, uint32_t sh, int32_t t_first, uint32_t rts, bool up2) {
  const uint32_t yrow = lzm_group_row(lane, ya, yb, g, rts);
  const MTap m = merge_taps(quantize_ltap(make_ltap(yrow, scy)), sh);
  int32_t tmax = ltap_i0(lzm_group_row(lane | 15u, ya, yb, g, rts), scy) + 3;
  tmax = (tmax < 0 ? 0 : (tmax > (int32_t)sh - 1 ? (int32_t)sh - 1 : tmax)) >> 4;
  u32x4* z = reinterpret_cast<u32x4*>(wm);
#pragma unroll
  for (int i = 0; i < (int)(kLzmWmBytes / 1024); i++) z[i * 64 + lane] = u32x4{0, 0, 0, 0};
  uint8_t* const cell = wm + ((lane >> 4) * 2 * 64 + (lane & 15)) * 16;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (m.pos[k] < 0) continue;
    const uint32_t pos = (uint32_t)m.pos[k], p = up2 ? ((pos >> 4) + 1u - (uint32_t)tmax) & 1u : ((pos >> 4) - (uint32_t)t_first) & 3;
    int32_t hi, lo;
    split_i8(m.q[k], hi, lo);
    uint8_t* const a = cell + (p >> 1) * 1024 + ((pos >> 2) & 3) * 256 + 8 * (p & 1) + 2 * (pos & 3);
    *reinterpret_cast<uint16_t*>(a) = (uint16_t)(((uint32_t)hi & 0xffu) | (((uint32_t)lo & 0xffu) << 8));
  }
}

// table builders: one wave per strip / per (group, band).  Column table of a plane row: [strip][plane hi | lo][K chunk][tile][lane][16 B];
// row table of a (plane height, band height): [band][group][kLzmWmBytes]
template <int NT, int KC>
__global__ __launch_bounds__(64) void k_lzm_build_cols(uint32_t ch, uint32_t sw, uint32_t dw, float scx, u32x4* __restrict__ tab) {
  __shared__ u32x4 scratch[2 * kLzmB1Chunk * 64];
  const uint32_t lane = threadIdx.x, dwb = dw * ch, ob0 = blockIdx.x * (16u * NT);
  v4i b1h[NT * KC], b1l[NT * KC];
  uint8_t* const lds = reinterpret_cast<uint8_t*>(scratch);
  switch (ch) {
    case 1: lzm_col_operands<1, NT, KC>(lds, lane, ob0, dwb, sw, scx, b1h, b1l); break;
    case 2: lzm_col_operands<2, NT, KC>(lds, lane, ob0, dwb, sw, scx, b1h, b1l); break;
    default: lzm_col_operands<3, NT, KC>(lds, lane, ob0, dwb, sw, scx, b1h, b1l); break;
  }
  u32x4* const out = tab + (size_t)blockIdx.x * (NT * KC * 128u);
#pragma unroll
  for (int i = 0; i < NT * KC; i++) {
    out[i * 64 + lane] = __builtin_bit_cast(u32x4, b1h[i]);
    out[(NT * KC + i) * 64 + lane] = __builtin_bit_cast(u32x4, b1l[i]);
  }
}
__global__ __launch_bounds__(64) void k_lzm_build_rows(uint32_t sh, uint32_t dh, float scy, uint32_t R, uint32_t rts, uint32_t up2, u32x4* __restrict__ tab) {
  __shared__ u32x4 wm[kLzmWmBytes / 16];
  const uint32_t lane = threadIdx.x, g = blockIdx.x, ya = blockIdx.y * R;
  if (ya >= dh) return;
  const uint32_t yb = ya + R - 1 < dh - 1 ? ya + R - 1 : dh - 1;
  if (g > (yb - ya) >> (rts + 2u)) return;
  lzm_row_group(reinterpret_cast<uint8_t*>(wm), lane, ya, yb, g, scy, sh, lzm_band_first_tile(ya, scy, sh), rts, up2 != 0);
  wave_lds_sync();
  u32x4* const out = tab + (size_t)(blockIdx.y * gridDim.x + g) * (kLzmWmBytes / 16);
#pragma unroll
  for (int i = 0; i < (int)(kLzmWmBytes / 1024); i++) out[i * 64 + lane] = wm[i * 64 + lane];
}

template <int CH, int NT, int PF, int KC, bool UP2>
VPF_DEV void LanczosMfmaTask<CH, NT, PF, KC, UP2>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp,
                                              const PlaneGeom& G, uint32_t bx, uint32_t by, const u32x4* __restrict__ ctab, const u32x4* __restrict__ rtab) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh, R = G.a1;
  const uint32_t rts = G.a3, rt = 1u << rts;  // destination rows per 16-row tile: 16, or 8 (half tiles: vertical factors of ~2.9 .. 6, see lzm_group_row)
  constexpr uint32_t P = lzm_pitch_of(PF);  // LDS pitch of a staged row: the variant's capacity (64 PF bytes) + 32, a compile-time constant (launcher: G.a0 == P)
  const float scx = G.scx, scy = G.scy;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t dwb = dw * CH, ob0 = (bx * 4 + wv) * (16u * NT), ya = by * R;
  if (bx * 4 * (16u * NT) >= dwb || ya >= dh) return;  // workgroup-uniform: this plane is narrower / shorter than the launch grid
  const uint32_t yb = ya + R - 1 < dh - 1 ? ya + R - 1 : dh - 1;
  uint8_t* const wmb = reinterpret_cast<uint8_t*>(dyn_strip);  // workgroup-shared: two groups x [4 tiles][2 planes][64 lanes][16 B] (first: an LDS-DMA target's address travels in M0)
  uint8_t* const lds = wmb + 2u * kLzmWmBytes + (size_t)wv * G.a2;
  uint8_t* const stage = lds;                         // [16 rows][P]
  uint8_t* const ot = lds + 16u * P;                  // [16 rows][lzm_out_pitch]
  constexpr uint32_t PO = lzm_out_pitch(NT);

  // ---- row weights, shared by the workgroup.  Its four waves own four neighbouring strips of the SAME band: wave (G & 3) brings in group G
  // (the 64 destination rows from ya + 64 G) for all four — a copy of the shape's row table (rtab), or, without
  // a table, evaluated here — into buffer G & 1, one group ahead of its use; the waves meet at one barrier per group.
  const int32_t t_first = __builtin_amdgcn_readfirstlane(lzm_band_first_tile(ya, scy, sh));  // first source tile of the band
  const uint32_t ngroups = ((yb - ya) >> (rts + 2u)) + 1u;  // a group = four tiles
  auto produce = [&](uint32_t g) {
    uint8_t* const wm = wmb + (g & 1u) * kLzmWmBytes;
    if (rtab) {
      // a straight 8-KiB copy, global memory -> LDS: eight LDS-DMA instructions (lane l's 16 bytes land at M0 + offset + 16 l), no register
      // and no ds_write involved.  The compiler does not see these loads: the wave that issued them waits for them by hand (group_ready)
      // before the barrier that hands the group to the others.  M0 is the compiler's: saved, set and restored inside each statement.
      const u32x4* const t = rtab + (size_t)(by * ((R + (4u << rts) - 1u) >> (rts + 2u)) + g) * (kLzmWmBytes / 16);
      const uint32_t voff = 16u * lane, ldst = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(wm));
      static_assert(kLzmWmBytes == 8192, "two statements of four 1-KiB pieces");
#pragma unroll
      for (int h = 0; h < 2; h++) {
        uint32_t keep;
        // (the table address is workgroup-uniform; spelled through readfirstlane so that the "s" operand is a scalar register pair whatever
        // the register allocator made of the values it is computed from — a build with more live state had handed the asm a VGPR pair)
        const uint64_t ta = reinterpret_cast<uint64_t>(reinterpret_cast<const uint8_t*>(t) + 4096 * h);
        const uint64_t tsc = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ta >> 32)) << 32 | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ta);  // (the builtin returns int: widened unsigned)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %1\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:2048\n\tglobal_load_lds_dwordx4 %2, %1 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(tsc), "v"(voff), "s"(ldst + 4096u * h) : "memory");
      }
    } else {
      lzm_row_group(wm, lane, ya, yb, g, scy, sh, t_first, rts, UP2);
    }
  };
  // the wave that brought group g in by DMA: everything it has in flight must have landed before it enters the barrier in front of g's use
  auto group_ready = [&](uint32_t g) {
    if (rtab && wv == (g & 3u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // the step from group g to g + 1: everybody is done with g's buffer, g + 1 is complete; g + 2 goes where g was
  auto next_group = [&](uint32_t g) {
    group_ready(g + 1);
    __syncthreads();
    if (g + 2 < ngroups && wv == ((g + 2) & 3u)) produce(g + 2);
  };
  if (wv == 0) produce(0);
  if (wv == 1 && ngroups > 1) produce(1);
  if (ob0 >= dwb) {  // a wave without columns (the row's last workgroup): it still produces its groups and meets the others
    group_ready(0); group_ready(1);
    __syncthreads();
    for (uint32_t g = 0; g + 1 < ngroups; g++) next_group(g);
    return;
  }

  // ---- windows: ws_j = 16-B aligned source byte below the first tap of tile j's first pixel (tiles past the row end copy the last window)
  const uint32_t S0 = __builtin_amdgcn_readfirstlane(lzm_window<CH>(ob0, dwb, sw, scx));
  uint32_t wrel[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) wrel[j] = __builtin_amdgcn_readfirstlane(lzm_window<CH>(ob0 + 16u * j, dwb, sw, scx)) - S0;

  // ---- column weights -> pass-1 B operands: the strip's 2 NT operands from the shape's column table (ctab), or
  // evaluated here
  v4i b1h[NT * KC], b1l[NT * KC];  // [K chunk * NT + tile]
  if (ctab) {
    const u32x4* const t = ctab + (size_t)(bx * 4 + wv) * (NT * KC * 128u);
#pragma unroll
    for (int i = 0; i < NT * KC; i++) {
      b1h[i] = __builtin_bit_cast(v4i, t[i * 64 + lane]);
      b1l[i] = __builtin_bit_cast(v4i, t[(NT * KC + i) * 64 + lane]);
    }
  } else {
    lzm_col_operands<CH, NT, KC>(lds, lane, ob0, dwb, sw, scx, b1h, b1l);
  }

  // ---- the march
  // 16-B units of a staged row (<= 4 PF, <= P / 16: host), cut at the row's last unit: the last window of a strip at the right image
  // edge reaches past the row (those bytes carry no weight and are never loaded: the LDS keeps whatever it held)
  const uint32_t nq_win = (wrel[NT - 1] + 64u * KC) / 16u, nq_row = (sw * CH + 15u - S0) / 16u;
  const uint32_t nq = nq_win < nq_row ? nq_win : nq_row;
  // staging: lane -> (row, 16-B unit) of the tile, PF loads per lane.  Even PF: EIGHT consecutive lanes take eight consecutive units (128 B)
  // of one row — the eight lanes a ds_write_b128 serves per LDS cycle then cover all 32 banks once (four lanes per row at a pitch of
  // 32 mod 64 put two rows' units on the same banks: the 2-way conflicts the counters showed) — load k = (row half k / (PF / 2), unit
  // column k % (PF / 2)): rows lane >> 3 and + 8, units (lane & 7) + 8 c.  Odd PF (the 320-B rows of the 2x down-scales): four lanes per row.
  constexpr int HALF = PF % 2 == 0 ? PF / 2 : PF, LPR = PF % 2 == 0 ? 8 : 4;
  auto k_row = [&](int k) -> uint32_t { return LPR == 8 ? (lane >> 3) + 8u * (uint32_t)(k / HALF) : lane >> 2; };
  auto k_unit = [&](int k) -> uint32_t { return (lane & (LPR - 1)) + (uint32_t)LPR * (uint32_t)(k % HALF); };
  // the units a lane stages, clamped to the strip's last one instead of predicated (all loads issue back to back; a clamped duplicate is
  // loaded from the last unit's address and stored at its OWN place in the row, which the pitch has room for and no weight reaches).
  // voff[k] = the lane's byte offset inside a 16-row source tile.  The tile itself is a scalar step on the base of a BUFFER descriptor that
  // ends with the plane's last row: a fetch costs no vector arithmetic at all (it was 8 instructions per tile), and the rows of the
  // picture's last, partial tile (sh % 16 != 0) that lie below the picture are out of the buffer's range — they read as zeros instead of
  // faulting, and carry no weight (the row weights of clamped taps sit on the last real row)
  // Even PF: load k and load k + PF / 2 are the same unit of rows 8 apart — one vector offset serves both, the row step travels as the
  // buffer load's SCALAR offset (round 5: PF / 2 VGPRs instead of PF; LzMfma4k8 sat at 256 VGPRs + one spilled)
  constexpr int NV = LPR == 8 ? HALF : PF;
  uint32_t voff[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) voff[k] = mad24(k_row(k), sp, S0 + 16u * (k_unit(k) < nq ? k_unit(k) : nq - 1u));
  const uint32_t sp8 = 8u * sp;  // scalar
  // Source tiles travel global memory -> registers -> LDS, TWO tiles ahead of the arithmetic (two register sets: the march is unrolled
  // four deep, so "which set" is a compile-time constant).  One tile ahead left the waves waiting for HBM: with two waves per SIMD a
  // tile's arithmetic lasts ~1 us, less than a loaded chip's memory latency (SQ_WAIT_ANY was 42 % of the wave cycles).
  int32_t t_last;                                  // last source tile of the band (nothing past it is fetched)
  {
    int32_t r = ltap_i0(yb, scy) + 3;
    r = r < 0 ? 0 : (r > (int32_t)sh - 1 ? (int32_t)sh - 1 : r);
    t_last = __builtin_amdgcn_readfirstlane(r >> 4);
  }
  const uint32_t plane_bytes = sh * sp;  // < 2^32: launcher
  // (every fetch issues exactly PF loads, predicated on nothing — units past the strip are clamped duplicates, tiles past the band's last
  // re-read the last — so that the compiler can count: the wait in front of a commit is vmcnt(PF), not vmcnt(0))
  // (the ring of two — three waves per SIMD, 168 registers — keeps ONE tile in flight: the third wave covers what the second set did)
  constexpr bool LEAN = LanczosMfmaTask<CH, NT, PF, KC, UP2>::kLean;
  constexpr int PD = LEAN ? 1 : 2;
  u32x4 pf[PD][PF];
  auto fetch = [&](int32_t T, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    const uint32_t toff = (uint32_t)(T < t_last ? T : t_last) * 16u * sp;  // scalar
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) + toff, 0, plane_bytes - toff, 0x00020000);
#pragma unroll
    for (int k = 0; k < PF; k++) pf[SET][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[k % NV], LPR == 8 && k >= HALF ? (int)sp8 : 0, 0));
  };
  v4i ring[NT][UP2 ? 1 : 2];  // per N-tile: two K chunks x (two source tiles x two dwords of (zl, zh) byte pairs); UP2: one chunk, (previous tile, this tile)
  static_assert(!UP2 || KC == 1, "the ring of two: one-chunk windows only");
  // where pass 1 of the tile in slot SLOT puts its two dwords
  auto ring_put = [&](int j, auto slot_tag, int32_t v0, int32_t v1) {
    constexpr int SLOT = decltype(slot_tag)::value;
    if constexpr (UP2) {
      ring[j][0][0] = ring[j][0][2]; ring[j][0][1] = ring[j][0][3];
      ring[j][0][2] = v0; ring[j][0][3] = v1;
    } else {
      ring[j][SLOT >> 1][2 * (SLOT & 1)] = v0; ring[j][SLOT >> 1][2 * (SLOT & 1) + 1] = v1;
    }
  };
#pragma unroll
  for (int j = 0; j < NT; j++) {
    ring[j][0] = v4i{0, 0, 0, 0};
    if constexpr (!UP2) ring[j][1] = v4i{0, 0, 0, 0};
  }
  const v4i c128 = {128, 128, 128, 128};
  // A operand of pass 1: lane (i, g) -> row i, bytes 16 g .. of tile j's window (one address register per tile: the kernel is issue-bound,
  // an add per read is 7 % of pass 1)
  // (the ring of two: one register + the tile's scalar window offset, added at the read — seven registers for one VALU instruction per read)
  const uint8_t* aptr[LEAN ? 1 : NT];
#pragma unroll
  for (int j = 0; j < (LEAN ? 1 : NT); j++) aptr[j] = stage + (lane & 15) * P + 16u * (lane >> 4) + wrel[j];
  auto a_of = [&](int j) -> const uint8_t* {
    if constexpr (LEAN) {
      uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(aptr[0]);
      asm volatile("" : "+v"(a));  // (not to be hoisted out of the march into eight registers again)
      return reinterpret_cast<const uint8_t*>(dyn_strip) + (a - (uint32_t)reinterpret_cast<uintptr_t>(dyn_strip)) + (wrel[j] - wrel[0]);
    } else {
      return aptr[j];
    }
  };
  uint8_t* const sdst = stage + (LPR == 8 ? (lane >> 3) : (lane >> 2)) * P + 16u * (lane & (LPR - 1));  // + 8 P per row half, + 16 LPR per unit column: immediates

  // pass 1 of source tile T into ring slot SLOT = (T - t_first) & 3 (a compile-time constant: the march below is unrolled four deep so
  // that the ring never moves in the register file)
  auto pass1 = [&](int32_t T, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int k = 0; k < PF; k++)
      *reinterpret_cast<u32x4*>(sdst + (LPR == 8 ? (uint32_t)(k / HALF) * 8u * P : 0u) + (uint32_t)(k % HALF) * (16u * LPR)) =
          pf[SLOT & (PD - 1)][k] ^ u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    fetch(T + PD, std::integral_constant<int, SLOT & (PD - 1)>{});
    wave_lds_sync();
    if constexpr (KC == 1) {
      // the A operands of the first four N-tiles are requested together, before the first is used (left alone the compiler keeps two reads in
      // flight and the wave waits four times per tile); the other four are requested one by one into the registers the MFMAs free
      constexpr int AD = LEAN ? 2 : 4;  // A operands in flight (the ring of two runs three waves per SIMD: its register budget is 168)
      v4i av[AD];
#pragma unroll
      for (int j = 0; j < AD; j++) av[j] = *reinterpret_cast<const v4i*>(a_of(j));
      if constexpr (AD == 4) asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
      else asm volatile("" : "+v"(av[0]), "+v"(av[1]));  // (an empty asm that "uses" them all: nothing may sink below it)
      // software pipeline over the N-tiles, interleaved at instruction level: the matrix pipe takes a new MFMA every 16 cycles and a wave issues
      // in order, so two MFMAs back to back park the wave for 12 cycles and the eight VALU instructions behind them then run with the pipe
      // idle.  Order per tile: HI of tile j + 1 | the four shift-adds of tile j (16 cycles: the pipe's own time) | LO of tile j + 1 | the two
      // v_perm_b32 + two xor of tile j.  (Left alone the compiler gives every tile the same result registers: MFMA, MFMA, wait for the pipe,
      // unpack, next MFMA — the wave idles through every MFMA latency.)
      v4i hi[2], lo[2];
      hi[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[0], b1h[0], c128, 0, 0, 0);
      lo[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[0], b1l[0], c128, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; j++) {
        if (j + 1 < NT) {
          hi[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[(j + 1) & (AD - 1)], b1h[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) h[r] = ((uint32_t)hi[j & 1][r] << 8) + (uint32_t)lo[j & 1][r];
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NT) {
          lo[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[(j + 1) & (AD - 1)], b1l[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j + AD < NT) {
          av[j & (AD - 1)] = *reinterpret_cast<const v4i*>(a_of(j + AD));
        }
        // bits 8 .. 23 of h'' = z + 128 (z = Hr - 8192): one v_perm_b32 per row pair packs two of them, the xor turns each low byte into the
        // signed zl (z = 256 zh + zl) — the ring holds (zl, zh) byte pairs, which is the K-slot order of pass 2's operands
        ring_put(j, slot_tag, (int32_t)(__builtin_amdgcn_perm(h[1], h[0], 0x06050201u) ^ 0x00800080u), (int32_t)(__builtin_amdgcn_perm(h[3], h[2], 0x06050201u) ^ 0x00800080u));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // KC K chunks per window: every product is KC chained MFMAs (chunk c accumulates onto chunk c - 1), 2 KC per N-tile against the same
      // eight VALU instructions — pass 1 is matrix-pipe-bound here.  All chunks of all A operands are requested up front (NT KC <= 8 reads);
      // per tile: the first-chunk HI of tile j + 1 | the four shift-adds of tile j | first-chunk LO | perm / xor of tile j | the other chunks
      static_assert(KC >= 2 && NT * KC <= 8, "multi-chunk windows: 4-tile strips with two chunks, 2-tile strips with three");
      v4i ac[NT][KC];
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int c = 0; c < KC; c++) ac[j][c] = *reinterpret_cast<const v4i*>(a_of(j) + 64 * c);
      v4i hi[2], lo[2];
      hi[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[0][0], b1h[0], c128, 0, 0, 0);
      lo[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[0][0], b1l[0], c128, 0, 0, 0);
#pragma unroll
      for (int c = 1; c < KC; c++) {
        hi[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[0][c], b1h[c * NT], hi[0], 0, 0, 0);
        lo[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[0][c], b1l[c * NT], lo[0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int n = (j + 1) & 1;
        if (j + 1 < NT) {
          hi[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[j + 1][0], b1h[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) h[r] = ((uint32_t)hi[j & 1][r] << 8) + (uint32_t)lo[j & 1][r];
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NT) {
          lo[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[j + 1][0], b1l[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        ring_put(j, slot_tag, (int32_t)(__builtin_amdgcn_perm(h[1], h[0], 0x06050201u) ^ 0x00800080u), (int32_t)(__builtin_amdgcn_perm(h[3], h[2], 0x06050201u) ^ 0x00800080u));
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NT) {
#pragma unroll
          for (int c = 1; c < KC; c++) {
            hi[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[j + 1][c], b1h[c * NT + j + 1], hi[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            lo[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[j + 1][c], b1l[c * NT + j + 1], lo[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    wave_lds_sync();  // the next tile's staging stores must not pass these reads
  };

  // pass 2 + store of one destination tile: rows y0 .. y0 + 15, weight operands of tile t of the current group
  const v4i cy = {(1 << 19) + (1 << 11), (1 << 19) + (1 << 11), (1 << 19) + (1 << 11), (1 << 19) + (1 << 11)};  // 8192 * 16384 / 256 + the rounding half
  const v4i czero = {0, 0, 0, 0};
  constexpr uint32_t LOGNT = NT == 8 ? 3 : NT == 4 ? 2 : 1, RPI = 64 / NT;  // read-back: lane -> (row lane >> LOGNT (+ RPI per pass), unit lane & (NT - 1))
  uint8_t* const owr = ot + (lane & 15) * PO + 4u * (lane >> 4);      // lane (y, g') writes bytes 4 g' .. 4 g' + 3 of every tile of row y
  const uint8_t* const ord = ot + (lane >> LOGNT) * PO + 16u * (lane & (NT - 1));
  const uint32_t ob = ob0 + 16u * (lane & (NT - 1));                   // first destination byte of the unit this lane stores
  const bool ofull = ob + 16u <= dwb, opart = !ofull && ob < dwb;
  const uint32_t obase = mad24(lane >> LOGNT, dp, ob);  // 32-bit offsets on the plane's scalar base (planes stay below 4 GiB: host)
  auto emit = [&](const uint8_t* wm, uint32_t t, uint32_t y0) {
    constexpr int C2 = 0;
    const v4i by0 = *reinterpret_cast<const v4i*>(wm + ((t * 2 + 0) * 64 + lane) * 16);
    v4i by1 = by0;
    if constexpr (!UP2) by1 = *reinterpret_cast<const v4i*>(wm + ((t * 2 + 1) * 64 + lane) * 16);
    // qh moves from the zl slot to the zh slot, the zl slots become 0: a left shift by 8 inside every 16-bit half (one v_pk_lshlrev_b16 per dword)
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v4i bx0 = __builtin_bit_cast(v4i, __builtin_bit_cast(v8s, by0) << 8), bx1 = __builtin_bit_cast(v4i, __builtin_bit_cast(v8s, by1) << 8);
    // software pipeline like pass 1, interleaved at instruction level: the four MFMAs of tile j + 1 (64 cycles of the matrix pipe) go out one
    // at a time between the ten VALU instructions that shift, clamp and pack tile j, instead of four in a row (the wave parked for 36
    // cycles) and the VALU work behind them.  The VALU work trails the MFMAs by most of a tile — the combine of tile j reads X, Y of tile j
    // three MFMA slots after Y was issued, its pack runs in tile j + 1's first two slots — so no VALU instruction waits out the matrix
    // pipe's result latency in s_nops
    v4i x[2], y[2];
    uint32_t pa = 0, pb = 0, w[4] = {0, 0, 0, 0};
    if constexpr (UP2) {
      // one chunk: X and Y of tile j + 1 go out around the pack of tile j - 1 and the combine of tile j
      x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][C2], bx0, czero, 0, 0, 0);
      y[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][C2], by0, cy, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const bool more = j + 1 < NT;
        if (more) { x[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][C2], bx0, czero, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        if (j > 0) {
          shift12_sat_pack4_a(w[0], w[1], w[2], pa, pb);
          *reinterpret_cast<uint32_t*>(owr + 16u * (j - 1)) = shift12_sat_pack4_b(pa, pb, w[3]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) { y[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][C2], by0, cy, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int r = 0; r < 4; r++) w[r] = ((uint32_t)x[j & 1][r] << 8) + (uint32_t)y[j & 1][r];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    {
      v4i t = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][0], bx0, czero, 0, 0, 0);
      v4i u = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][0], by0, cy, 0, 0, 0);
      x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][1], bx1, t, 0, 0, 0);
      y[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][1], by1, u, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const bool more = j + 1 < NT;
      v4i t, u;
      if (more) { t = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][0], bx0, czero, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      if (j > 0) { shift12_sat_pack4_a(w[0], w[1], w[2], pa, pb); __builtin_amdgcn_sched_barrier(0); }
      if (more) { u = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][0], by0, cy, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      if (j > 0) {
        *reinterpret_cast<uint32_t*>(owr + 16u * (j - 1)) = shift12_sat_pack4_b(pa, pb, w[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) { x[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][1], bx1, t, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      w[0] = ((uint32_t)x[j & 1][0] << 8) + (uint32_t)y[j & 1][0];  // V / 256 + 2^11 (Q12): the byte is w >> 12, clamped
      w[1] = ((uint32_t)x[j & 1][1] << 8) + (uint32_t)y[j & 1][1];
      __builtin_amdgcn_sched_barrier(0);
      if (more) { y[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][1], by1, u, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      w[2] = ((uint32_t)x[j & 1][2] << 8) + (uint32_t)y[j & 1][2];
      w[3] = ((uint32_t)x[j & 1][3] << 8) + (uint32_t)y[j & 1][3];
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    shift12_sat_pack4_a(w[0], w[1], w[2], pa, pb);
    *reinterpret_cast<uint32_t*>(owr + 16u * (NT - 1)) = shift12_sat_pack4_b(pa, pb, w[3]);
    wave_lds_sync();
    const uint32_t orow = mad24(y0, dp, obase);
#pragma unroll
    for (int it = 0; it < (NT + 3) / 4; it++) {
      const uint32_t rl = (lane >> LOGNT) + RPI * it, y = y0 + rl;  // row of the tile (2-tile strips: lanes 32 .. 63 have none)
      if (rl < rt && y <= yb) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ord + RPI * it * PO);
        uint8_t* const out = dst + (orow + (uint32_t)(RPI * it) * dp);
        if (ofull) {
          stg<true, u32x4>(out, v);
        } else if (opart) {  // the row's last, partial unit
          const uint32_t nb = dwb - ob;
          for (uint32_t i = 0; i < nb; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
        }
      }
    }
    wave_lds_sync();
  };

  // the last source tile each row of the current group needs (lane = row), for the emit test below
  auto group_tmax = [&](uint32_t g) -> int32_t {
    const uint32_t yrow = lzm_group_row(lane, ya, yb, g, rts);
    int32_t r = ltap_i0(yrow, scy) + 3;
    r = r < 0 ? 0 : (r > (int32_t)sh - 1 ? (int32_t)sh - 1 : r);
    return r >> 4;
  };
  uint32_t grp = 0, y_next = ya;  // current weight group / first row of the next destination tile to emit
  int32_t tmax_l = group_tmax(0);
  int32_t T = t_first;
  fetch(T, std::integral_constant<int, 0>{});
  if constexpr (PD == 2) fetch(T + 1, std::integral_constant<int, 1>{});
  group_ready(0); group_ready(1);
  __syncthreads();  // groups 0 and 1 are in LDS
  VPF_WAVE_MARK(0);  // (lab builds: setup done — row groups in LDS, column operands and the first two source tiles requested)
  // one step: the next source tile, then every destination tile whose last source tile it was
#define VPF_LZM_STEP(S)                                                                                              \
  pass1(T, std::integral_constant<int, S>{});                                                                       \
  if (T == t_first) VPF_WAVE_MARK(1); /* the first source tile has arrived and is through pass 1 */                 \
  T++;                                                                                                              \
  for (;;) {                                                                                                        \
    const uint32_t tl = ((y_next - ya) >> rts) & 3u;                                                                \
    if (__builtin_amdgcn_readlane(tmax_l, 16 * tl + rt - 1) >= T) break;                                            \
    emit(wmb + (grp & 1u) * kLzmWmBytes, tl, y_next);                                                               \
    if (y_next == ya) VPF_WAVE_MARK(2); /* the first destination tile is stored */                                  \
    y_next += rt;                                                                                                   \
    if (y_next > yb) return;                                                                                        \
    if (tl == 3) { next_group(grp); grp++; tmax_l = group_tmax(grp); }                                              \
  }
  for (;;) {
    VPF_LZM_STEP(0) VPF_LZM_STEP(1) VPF_LZM_STEP(2) VPF_LZM_STEP(3)
  }
#undef VPF_LZM_STEP
}

#ifdef VPF_LAB_FORMS  // measured and not selected by any policy: built into tools/lab/libvpfhip_forms.so only (knob VPF_TUNE_RESIZE_MFMA | 0x20000)
// ------------------------------------------------------------------------------------------------------------------------------------
// The same filter with the two passes on DIFFERENT waves (round 4).  LanczosMfmaTask keeps the column operands (64 VGPRs), the ring (64) and
// the prefetch sets in one wave: 230-250 registers, two waves per SIMD — and two in-order waves leave the SIMD's issue port 40 % idle
// (DESIGN.md 4.2).  Here a workgroup of four waves owns TWO neighbouring strips of a band: waves 0, 1 run pass 1 of strip 0, 1 (column
// operands + staging: ~150 registers), waves 2, 3 run pass 2 (ring + row weights: ~135), so three workgroups fit a CU and every SIMD holds
// three waves of mixed roles.  A pass-1 wave hands each source tile's (zl, zh) pairs — 16 dwords per lane — to its partner through a QUEUE
// of kLzpQueue 4-KiB LDS tiles guarded by two sequence words per slot (ready / freed): pass 2 is bursty (0, 1 or 2 destination tiles per
// source tile), pass 1 is not, and a first form that met at one workgroup barrier per source tile lost more in lockstep than the third wave
// brought (profiles/r04_lanczos_pair_sweep_barrier.txt).  No barrier after the start: the pass-2 waves read their row weights straight
// from the shape's table (L2-resident, two 16-B loads per lane and destination tile, requested a tile ahead), so there are no shared
// row-weight groups either; a launch without tables takes LanczosMfmaTask.
// Same arithmetic on the same operands -> the same bytes as LanczosMfmaTask and the oracle.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kLzpQueue = 4;
template <int CH, int PF>
struct LanczosPairTask {
  static constexpr int NT = 8;
  static constexpr int kThreads = 256;
  static constexpr int kGroupsPerCu = 3;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by,
                          const u32x4* __restrict__ ctab, const u32x4* __restrict__ rtab);
};
// a sequence word in LDS: written by one wave, polled by its partner.  Plain DS instructions on the word's LDS address (a generic pointer
// would make these flat accesses, which also wait for every global load in flight); no s_waitcnt around the store: a CU's LDS serves one
// wave's requests in the order they were issued, so the payload's ds_writes are in front of the word and the payload's ds_reads — issued
// after the poll has returned — behind it.  The "memory" clobbers keep the compiler from moving its own LDS accesses across.
VPF_DEV void lzp_post(uint32_t lds_addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(v) : "memory");
}
VPF_DEV void lzp_await(uint32_t lds_addr, uint32_t v) {
  for (uint32_t spin = 0; spin < (1u << 24); spin++) {  // (bounded: a protocol error must show as wrong pixels in a test, never as a hung GPU)
    uint32_t got;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(lds_addr) : "memory");
    if ((int32_t)(__builtin_amdgcn_readfirstlane(got) - v) >= 0) break;
    __builtin_amdgcn_s_sleep(2);
  }
}
template <int CH, int PF>
VPF_DEV void LanczosPairTask<CH, PF>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                          uint32_t bx, uint32_t by, const u32x4* __restrict__ ctab, const u32x4* __restrict__ rtab) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh, R = G.a1;
  constexpr uint32_t P = lzm_pitch_of(PF), PO = lzm_out_pitch(NT);
  const float scx = G.scx, scy = G.scy;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool consumer = wv >= 2;                 // waves 2, 3: pass 2 of strip 0, 1
  const uint32_t pr = wv & 1u;                   // the pair (= strip of the workgroup) this wave belongs to
  const uint32_t dwb = dw * CH, strip = bx * 2 + pr, ob0 = strip * (16u * NT), ya = by * R;
  // LDS: per pair the staged source tile | kLzpQueue hand-over tiles | out tile | 2 x kLzpQueue sequence words
  constexpr uint32_t HX_B = 8u * 64u * 8u;       // hand-over tile: [N-tile][lane][two dwords of (zl, zh) pairs]
  uint8_t* const pbase = reinterpret_cast<uint8_t*>(dyn_strip) + (size_t)pr * lzm_pair_lds(PF);
  uint8_t* const stage = pbase;                  // [16 rows][P]
  uint8_t* const hx = pbase + 16u * P;           // [kLzpQueue][HX_B]
  uint8_t* const ot = hx + kLzpQueue * HX_B;     // [16 rows][PO]
  const uint32_t seq = (uint32_t)reinterpret_cast<uintptr_t>(ot + 16u * PO);  // LDS address of ready[kLzpQueue] | freed[kLzpQueue] (4 B each)
  if (threadIdx.x < 4u * kLzpQueue) {            // both pairs' words start at 0 (workgroup-wide, before anybody leaves)
    volatile uint32_t* const z = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint8_t*>(dyn_strip) + (size_t)(threadIdx.x / (2u * kLzpQueue)) * lzm_pair_lds(PF) + 16u * P +
                                                                       kLzpQueue * HX_B + 16u * PO);
    z[threadIdx.x % (2u * kLzpQueue)] = 0;
  }
  __syncthreads();
  if (bx * 2 * (16u * NT) >= dwb || ya >= dh || ob0 >= dwb) return;  // a plane narrower / shorter than the launch grid; a pair without columns
  const uint32_t yb = ya + R - 1 < dh - 1 ? ya + R - 1 : dh - 1;
  const int32_t t_first = __builtin_amdgcn_readfirstlane(lzm_band_first_tile(ya, scy, sh));
  int32_t t_last;
  {
    int32_t r = ltap_i0(yb, scy) + 3;
    r = r < 0 ? 0 : (r > (int32_t)sh - 1 ? (int32_t)sh - 1 : r);
    t_last = __builtin_amdgcn_readfirstlane(r >> 4);
  }
  const uint32_t ntiles = (uint32_t)(t_last - t_first + 1);  // source tiles of the band: hand-overs 1 .. ntiles

  if (!consumer) {
    // ================================================================== pass-1 wave
    const uint32_t S0 = __builtin_amdgcn_readfirstlane(lzm_window<CH>(ob0, dwb, sw, scx));
    uint32_t wrel[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) wrel[j] = __builtin_amdgcn_readfirstlane(lzm_window<CH>(ob0 + 16u * j, dwb, sw, scx)) - S0;
    v4i b1h[NT], b1l[NT];
    {
      const u32x4* const t = ctab + (size_t)strip * (NT * 128u);
#pragma unroll
      for (int j = 0; j < NT; j++) {
        b1h[j] = __builtin_bit_cast(v4i, t[j * 64 + lane]);
        b1l[j] = __builtin_bit_cast(v4i, t[(NT + j) * 64 + lane]);
      }
    }
    const uint32_t nq_win = (wrel[NT - 1] + 64u) / 16u, nq_row = (sw * CH + 15u - S0) / 16u;
    const uint32_t nq = nq_win < nq_row ? nq_win : nq_row;
    constexpr int HALF = PF % 2 == 0 ? PF / 2 : PF, LPR = PF % 2 == 0 ? 8 : 4;
    auto k_row = [&](int k) -> uint32_t { return LPR == 8 ? (lane >> 3) + 8u * (uint32_t)(k / HALF) : lane >> 2; };
    auto k_unit = [&](int k) -> uint32_t { return (lane & (LPR - 1)) + (uint32_t)LPR * (uint32_t)(k % HALF); };
    uint32_t voff[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) voff[k] = mad24(k_row(k), sp, S0 + 16u * (k_unit(k) < nq ? k_unit(k) : nq - 1u));
    const uint32_t plane_bytes = sh * sp;
    u32x4 pf[2][PF];
    auto fetch = [&](int32_t T, auto set_tag) {
      constexpr int SET = decltype(set_tag)::value;
      const uint32_t toff = (uint32_t)(T < t_last ? T : t_last) * 16u * sp;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) + toff, 0, plane_bytes - toff, 0x00020000);
#pragma unroll
      for (int k = 0; k < PF; k++) pf[SET][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[k], 0, 0));
    };
    const v4i c128 = {128, 128, 128, 128};
    const uint8_t* aptr[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) aptr[j] = stage + (lane & 15) * P + 16u * (lane >> 4) + wrel[j];
    uint8_t* const sdst = stage + (LPR == 8 ? (lane >> 3) : (lane >> 2)) * P + 16u * (lane & (LPR - 1));
    uint8_t* const hxw = hx + lane * 8u;
    // tile number k (0-based) of the band: staged from register set k & 1, handed over in queue slot k % kLzpQueue
    auto pass1 = [&](uint32_t k, auto par_tag) {
      constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
      for (int q = 0; q < PF; q++)
        *reinterpret_cast<u32x4*>(sdst + (LPR == 8 ? (uint32_t)(q / HALF) * 8u * P : 0u) + (uint32_t)(q % HALF) * (16u * LPR)) =
            pf[PAR][q] ^ u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
      fetch(t_first + (int32_t)k + 2, std::integral_constant<int, PAR>{});
      wave_lds_sync();
      v4i av[4];
#pragma unroll
      for (int j = 0; j < 4; j++) av[j] = *reinterpret_cast<const v4i*>(aptr[j]);
      asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
      const uint32_t slot = k % kLzpQueue;
      if (k >= kLzpQueue) lzp_await(seq + 4u * (kLzpQueue + slot), k - kLzpQueue + 1u);  // the partner has taken hand-over k - kLzpQueue + 1 out of this slot
      uint8_t* const hxs = hxw + slot * HX_B;
      v4i hi[2], lo[2];
      hi[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[0], b1h[0], c128, 0, 0, 0);
      lo[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[0], b1l[0], c128, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; j++) {
        if (j + 1 < NT) {
          hi[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[(j + 1) & 3], b1h[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) h[r] = ((uint32_t)hi[j & 1][r] << 8) + (uint32_t)lo[j & 1][r];
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NT) {
          lo[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[(j + 1) & 3], b1l[j + 1], c128, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 4 < NT) av[j & 3] = *reinterpret_cast<const v4i*>(aptr[j + 4]);
        const u32x2 d = {__builtin_amdgcn_perm(h[1], h[0], 0x06050201u) ^ 0x00800080u, __builtin_amdgcn_perm(h[3], h[2], 0x06050201u) ^ 0x00800080u};
        *reinterpret_cast<u32x2*>(hxs + j * 512u) = d;
        __builtin_amdgcn_sched_barrier(0);
      }
      lzp_post(seq + 4u * slot, k + 1u);
    };
    fetch(t_first, std::integral_constant<int, 0>{});
    fetch(t_first + 1, std::integral_constant<int, 1>{});
    for (uint32_t k = 0; k < ntiles; k += 2) {
      pass1(k, std::integral_constant<int, 0>{});
      if (k + 1 < ntiles) pass1(k + 1, std::integral_constant<int, 1>{});
    }
    return;
  }

  // ==================================================================== pass-2 wave
  const uint32_t gpb = (R + 63u) / 64u;
  v4i ring[NT][2];
#pragma unroll
  for (int j = 0; j < NT; j++) { ring[j][0] = v4i{0, 0, 0, 0}; ring[j][1] = v4i{0, 0, 0, 0}; }
  const v4i cy = {(1 << 19) + (1 << 11), (1 << 19) + (1 << 11), (1 << 19) + (1 << 11), (1 << 19) + (1 << 11)};
  const v4i czero = {0, 0, 0, 0};
  constexpr uint32_t LOGNT = 3, RPI = 64 / NT;
  uint8_t* const owr = ot + (lane & 15) * PO + 4u * (lane >> 4);
  const uint8_t* const ord = ot + (lane >> LOGNT) * PO + 16u * (lane & (NT - 1));
  const uint32_t ob = ob0 + 16u * (lane & (NT - 1));
  const bool ofull = ob + 16u <= dwb, opart = !ofull && ob < dwb;
  const uint32_t obase = mad24(lane >> LOGNT, dp, ob);
  const uint8_t* const hxr = hx + lane * 8u;
  // row-weight operands of destination tile number n of the band (group n / 4, tile n % 4): the Y operand of the ring's two K chunks
  v4i byr[2];
  auto preload = [&](uint32_t n) {
    const u32x4* const t = rtab + (size_t)(by * gpb + (n >> 2)) * (kLzmWmBytes / 16) + (size_t)((n & 3u) * 2u) * 64u + lane;
    byr[0] = __builtin_bit_cast(v4i, t[0]);
    byr[1] = __builtin_bit_cast(v4i, t[64]);
  };
  auto emit = [&](uint32_t y0) {
    const v4i by0 = byr[0], by1 = byr[1];
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v4i bx0 = __builtin_bit_cast(v4i, __builtin_bit_cast(v8s, by0) << 8), bx1 = __builtin_bit_cast(v4i, __builtin_bit_cast(v8s, by1) << 8);
    v4i x[2], y[2];
    {
      v4i tt = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][0], bx0, czero, 0, 0, 0);
      v4i uu = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][0], by0, cy, 0, 0, 0);
      x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][1], bx1, tt, 0, 0, 0);
      y[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[0][1], by1, uu, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t pa = 0, pb = 0, w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const bool more = j + 1 < NT;
      v4i tt, uu;
      if (more) { tt = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][0], bx0, czero, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      if (j > 0) { shift12_sat_pack4_a(w[0], w[1], w[2], pa, pb); __builtin_amdgcn_sched_barrier(0); }
      if (more) { uu = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][0], by0, cy, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      if (j > 0) {
        *reinterpret_cast<uint32_t*>(owr + 16u * (j - 1)) = shift12_sat_pack4_b(pa, pb, w[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) { x[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][1], bx1, tt, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      w[0] = ((uint32_t)x[j & 1][0] << 8) + (uint32_t)y[j & 1][0];
      w[1] = ((uint32_t)x[j & 1][1] << 8) + (uint32_t)y[j & 1][1];
      __builtin_amdgcn_sched_barrier(0);
      if (more) { y[(j + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[j + 1][1], by1, uu, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
      w[2] = ((uint32_t)x[j & 1][2] << 8) + (uint32_t)y[j & 1][2];
      w[3] = ((uint32_t)x[j & 1][3] << 8) + (uint32_t)y[j & 1][3];
      __builtin_amdgcn_sched_barrier(0);
    }
    shift12_sat_pack4_a(w[0], w[1], w[2], pa, pb);
    *reinterpret_cast<uint32_t*>(owr + 16u * (NT - 1)) = shift12_sat_pack4_b(pa, pb, w[3]);
    wave_lds_sync();
    const uint32_t orow = mad24(y0, dp, obase);
#pragma unroll
    for (int it = 0; it < NT / 4; it++) {
      const uint32_t yy = y0 + (lane >> LOGNT) + RPI * it;
      if (yy <= yb) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ord + RPI * it * PO);
        uint8_t* const out = dst + (orow + (uint32_t)(RPI * it) * dp);
        if (ofull) {
          stg<true, u32x4>(out, v);
        } else if (opart) {
          const uint32_t nb = dwb - ob;
          for (uint32_t i = 0; i < nb; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
        }
      }
    }
    wave_lds_sync();
  };
  auto group_tmax = [&](uint32_t g) -> int32_t {
    const uint32_t yrow = ya + 64u * g + lane < yb ? ya + 64u * g + lane : yb;
    int32_t r = ltap_i0(yrow, scy) + 3;
    r = r < 0 ? 0 : (r > (int32_t)sh - 1 ? (int32_t)sh - 1 : r);
    return r >> 4;
  };
  uint32_t n_next = 0, y_next = ya;               // number / first row of the next destination tile
  int32_t tmax_l = group_tmax(0);
  bool done = false;
  preload(0);
  // hand-over k + 1 (source tile t_first + k) into ring slot k & 3, then every destination tile it completes
  auto consume = [&](uint32_t k, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const uint32_t qs = k % kLzpQueue;
    lzp_await(seq + 4u * qs, k + 1u);
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const u32x2 d = *reinterpret_cast<const u32x2*>(hxr + qs * HX_B + j * 512u);
      ring[j][SLOT >> 1][2 * (SLOT & 1)] = (int32_t)d[0];
      ring[j][SLOT >> 1][2 * (SLOT & 1) + 1] = (int32_t)d[1];
    }
    asm volatile("" : "+v"(ring[0][SLOT >> 1]), "+v"(ring[NT - 1][SLOT >> 1]));  // (the reads are complete before the slot is given back)
    lzp_post(seq + 4u * (kLzpQueue + qs), k + 1u);
    const int32_t Tc = t_first + (int32_t)k;
    for (;;) {
      const uint32_t tl = n_next & 3u;
      if (__builtin_amdgcn_readlane(tmax_l, 16 * tl + 15) > Tc) break;
      emit(y_next);
      y_next += 16; n_next++;
      if (y_next > yb) { done = true; break; }
      if (tl == 3) tmax_l = group_tmax(n_next >> 2);
      preload(n_next);
    }
  };
  for (uint32_t k = 0; k < ntiles && !done;) {
    consume(k, std::integral_constant<int, 0>{}); if (++k >= ntiles || done) break;
    consume(k, std::integral_constant<int, 1>{}); if (++k >= ntiles || done) break;
    consume(k, std::integral_constant<int, 2>{}); if (++k >= ntiles || done) break;
    consume(k, std::integral_constant<int, 3>{}); ++k;
  }
}
#endif  // VPF_LAB_FORMS

template <int CH> struct LzMfma8 : LanczosMfmaTask<CH, 8, 4> {};   // strips of 8 tiles, staged rows of up to 256 B
template <int CH> struct LzMfma8n : LanczosMfmaTask<CH, 8, 2> {};  // ... of up to 128 B (up-scales)
template <int CH> struct LzMfma8w : LanczosMfmaTask<CH, 8, 5> {};  // ... of up to 320 B (2x down-scales)
template <int CH> struct LzMfma4 : LanczosMfmaTask<CH, 4, 4> {};
template <int CH> struct LzMfma4n : LanczosMfmaTask<CH, 4, 2> {};
template <int CH> struct LzMfma8u : LanczosMfmaTask<CH, 8, 2, 1, true> {};  // the ring of two (up-scales): one K chunk in pass 2
template <int CH> struct LzMfma4u : LanczosMfmaTask<CH, 4, 2, 1, true> {};
template <int CH> struct LzMfma8uw : LanczosMfmaTask<CH, 8, 4, 1, true> {};  // ... on 8-tile strips of 1.5 x up-scales (rows of up to 256 B)
template <int CH> struct LzMfma4k4 : LanczosMfmaTask<CH, 4, 4, 2> {};  // two-chunk windows (strong horizontal down-scales): staged rows of up to 256 B
template <int CH> struct LzMfma4k6 : LanczosMfmaTask<CH, 4, 6, 2> {};  // ... 384 B
template <int CH> struct LzMfma4k8 : LanczosMfmaTask<CH, 4, 8, 2> {};  // ... 512 B
template <int CH> struct LzMfma2k6 : LanczosMfmaTask<CH, 2, 6, 3> {};  // three-chunk windows, 2-tile strips (factors up to ~10): staged rows of up to 384 B
template <int CH> struct LzMfma2k8 : LanczosMfmaTask<CH, 2, 8, 3> {};  // ... 512 B
#ifdef VPF_LAB_FORMS
template <int CH> struct LzPair : LanczosPairTask<CH, 4> {};    // the two-role form: two 8-tile strips per workgroup, three workgroups per CU
template <int CH> struct LzPairN : LanczosPairTask<CH, 2> {};
template <int CH> struct LzPairW : LanczosPairTask<CH, 5> {};
#endif

// all planes of up to 32 frames in one dispatch (the k_planes_mp scheme of k_resize_common.h, with this family's register budget:
// two workgroups per CU, and the planes' weight tables)
struct LzmTableArgs {
  const u32x4* ctab[3];
  const u32x4* rtab[3];
};
template <template <int> class TaskCH, class BA = BatchArgs>  // BA: the frame table's size (<= 32 / <= 128 frames: vpf_internal.h)
__global__ __launch_bounds__(256, TaskCH<3>::kGroupsPerCu) void k_lanczos_mfma(const BA args, const PlaneTable T, const LzmTableArgs W) {
  VPF_WAVE_TIMER(3);
  const BlockId b = picture_order();  // XCD-aware numbering (k_resize_common.h): neighbouring strips and bands share one L2
  const uint32_t bx = b.x, by = b.y, bz = b.z;
  const FrameDesc& f = args.f[bz];
  const uint32_t pi = (uint32_t)(T.np > 1 && by >= T.by0[1]) + (uint32_t)(T.np > 2 && by >= T.by0[2]);
  const uint32_t k = T.k[pi], lby = by - T.by0[pi];
  switch (T.ch[pi]) {  // workgroup-uniform
    case 1: TaskCH<1>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], bx, lby, W.ctab[pi], W.rtab[pi]); break;
    case 2: TaskCH<2>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], bx, lby, W.ctab[pi], W.rtab[pi]); break;
    default: TaskCH<3>::run(f.s[k], f.sp[k], f.d[k], f.dp[k], T.g[pi], bx, lby, W.ctab[pi], W.rtab[pi]); break;
  }
}

// Two workgroups per CU share its 160 KB of LDS: up to 80 KB per workgroup, which is above the 64 KB a kernel gets without asking
// (a 2x down-scale with 8-tile strips stages 544-B rows: 76 KB).  hipFuncSetAttribute is per device and not a stream operation: done
// once per device and kernel, outside any capture-sensitive path (it neither allocates nor synchronises).
template <template <int> class TaskCH>
static bool lzm_big_lds_ok() {
  static std::atomic<uint64_t> done{0}, failed{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
  const uint64_t bit = 1ull << dev;
  if (done.load(std::memory_order_acquire) & bit) return !(failed.load(std::memory_order_acquire) & bit);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lanczos_mfma<TaskCH, BatchArgs>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLzmMaxLds);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lanczos_mfma<TaskCH, BatchArgsL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLzmMaxLds);  // (both frame-table sizes)
  if (e != hipSuccess) { (void)hipGetLastError(); failed.fetch_or(bit, std::memory_order_release); }
  done.fetch_or(bit, std::memory_order_release);
  return e == hipSuccess;
}

// the per-shape weight tables' bookkeeping (vpf_lzm_plan.h); VPF_HIP_LANCZOS_TABLE_KB shrinks the part of the arena that is handed out
// (0 = no tables at all): a test knob for the "arena full" path, read once
// HIP events behind the fallback arena's bookkeeping (vpf_lzm_plan.h: LzmSync).  The caller of launch_lanczos_mfma holds a DeviceGuard: the
// current device is the one the launch — and these events — belong to.
static std::atomic<u32x4*> g_lzm_arena_base[64];  // the static arena's address per device (lzm_arena_base)
struct LzmHipSync final : LzmSync {
  hipEvent_t canary[64] = {};
  void* record(const void* stream, int) override {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipEventRecord(e, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e); return nullptr; }
    return e;
  }
  bool done(void* ev) override {
    const hipError_t r = hipEventQuery((hipEvent_t)ev);
    if (r != hipSuccess && r != hipErrorNotReady) (void)hipGetLastError();
    return r != hipErrorNotReady;  // an event that can no longer be queried (device reset) holds nobody up
  }
  void wait(const void* stream, void* ev) override {
    if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) != hipSuccess) (void)hipGetLastError();
  }
  void destroy(void* ev) override {
    if (hipEventDestroy((hipEvent_t)ev) != hipSuccess) (void)hipGetLastError();
  }
  // a device reset frees the arena's contents (static device memory is re-initialised) and invalidates every event created before it: a
  // never-recorded canary event per device answers hipSuccess while its context lives.  A detected reset also forgets the arena's cached
  // address (the code object is loaded again: lzm_arena_base asks anew); the stale canary handle is dropped, never destroyed
  bool device_alive(int dev) override {
    if (canary[dev] && hipEventQuery(canary[dev]) == hipSuccess) return true;
    const bool first = canary[dev] == nullptr;
    (void)hipGetLastError();
    canary[dev] = nullptr;
    g_lzm_arena_base[dev].store(nullptr, std::memory_order_release);
    if (hipEventCreateWithFlags(&canary[dev], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); canary[dev] = nullptr; }
    return first;
  }
  // asked at eviction about a stream handle remembered from an earlier launch (vpf_lzm_plan.h): capturing -> hands off; a handle the
  // runtime no longer knows -> nothing to order
  int stream_state(const void* stream) override {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); return 2; }
    return cs != hipStreamCaptureStatusNone ? 1 : 0;
  }
};
// VPF_HIP_LANCZOS_TABLE_KB shrinks the part of the arena that is handed out (0 = no tables at all): a test knob, read once
static LzmTableCache& lzm_tables() {
  static LzmHipSync sync;
  static LzmTableCache cache([] {
    const char* e = std::getenv("VPF_HIP_LANCZOS_TABLE_KB");
    const uint64_t kb = e ? std::strtoull(e, nullptr, 10) : kLzmArenaBytes / 1024;
    return std::min<uint64_t>(kb * 1024, kLzmArenaBytes);
  }(), &sync);
  return cache;
}
static u32x4* lzm_arena_base(int dev) {  // the static arena's address on this device (hipGetSymbolAddress once per device)
  std::atomic<u32x4*>* const base = g_lzm_arena_base;
  u32x4* p = base[dev].load(std::memory_order_acquire);
  if (!p) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_lzm_arena)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    p = static_cast<u32x4*>(q);
    base[dev].store(p, std::memory_order_release);
  }
  return p;
}
// the workspace of the vpf_resize_ws / vpf_resize_batch_ws call this thread is inside (vpf_abi.hip sets it; nullptr otherwise)
thread_local vpf_workspace* t_lzm_workspace = nullptr;
void set_lanczos_workspace(vpf_workspace* ws) { t_lzm_workspace = ws; }
uint64_t lanczos_table_bytes_bound(int ch, uint32_t dw, uint32_t dh) { return lzm_table_bytes_bound(ch, dw, dh); }

bool launch_lanczos_mfma(hipStream_t st, int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a) {
  const int tune = tuning(VPF_TUNE_NV12_RGB_VARIANT), knob = tuning(VPF_TUNE_RESIZE_MFMA);
  const int forced = knob & 0xffff;           // 0 policy | 1 off | (nt << 8 | band rows / 16): measurement and test knob
  const bool tables = !(knob & 0x10000);      // | 0x10000: evaluate the weights in the kernel (the path a full arena takes)
#ifdef VPF_LAB_FORMS
  const bool pair = (knob & 0x20000) != 0;    // | 0x20000: the two-role form (LanczosPairTask): 8-tile strips only
#else
  constexpr bool pair = false;                // (the two-role form lives in the lab build: tools/lab/libvpfhip_forms.so)
#endif
  if (tune == 9 || tune == 40 || forced == 1) return false;
  if (njobs < 1 || njobs > 3 || !n || n > (uint32_t)kMaxBatch) return false;
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    if (j.sw >= (1u << 22) || j.sh >= (1u << 22) || j.dw >= (1u << 22) || j.dh >= (1u << 22)) return false;
    for (uint32_t i = 0; i < n; i++) {
      if ((((uintptr_t)a.f[i].s[j.k] | a.f[i].sp[j.k] | (uintptr_t)a.f[i].d[j.k] | a.f[i].dp[j.k]) & 15)) return false;
      // the kernel addresses a plane with 32-bit offsets built by 24-bit multiplies
      if (a.f[i].sp[j.k] >= (1u << 24) || a.f[i].dp[j.k] >= (1u << 24) || (uint64_t)j.sh * a.f[i].sp[j.k] >= (1ull << 32) || (uint64_t)j.dh * a.f[i].dp[j.k] >= (1ull << 32)) return false;
    }
    if (!lzm_shape(j.ch, j.sw, j.sh, j.dw, j.dh).rows_ok) return false;
  }
  LzmPlaneIn in[3];
  for (int p = 0; p < njobs; p++) in[p] = LzmPlaneIn{jobs[p].ch, jobs[p].sw, jobs[p].sh, jobs[p].dw, jobs[p].dh};
  // launch shape by the cost model of vpf_lzm_plan.h (| 0x80000: never the ring of two, the measurement and test knob)
  const LzmPlan plan = lzm_plan(njobs, in, n, pair ? ((8 << 8) | (forced & 0xff)) : forced, tables, !pair && !(knob & 0x80000));
  if (!plan.ok) return false;
  const int nt = plan.nt, kc = plan.kc;
  const uint32_t rts = (uint32_t)plan.rts;  // log2 of the destination rows per tile: 4, or 3 (half tiles)
  if (pair && rts != 4) return false;
  const uint32_t band_tiles = plan.band_tiles, span = plan.span, pitch = plan.pitch, wave_lds = plan.wave_lds;
  const bool narrow = span <= 2u * 64u;  // two staging loads per lane and tile cover the strip
  const bool up2 = plan.up2;  // the ring of two (LanczosMfmaTask<.., UP2>): narrow strips, every plane's destination tiles within two source tiles
  PlaneTable t{};
  LzmTableArgs wt{};
  t.np = (uint32_t)njobs;
  uint32_t gx = 0, gy = 0;
  int dev = 0;
  bool capturing = false;
  // where this launch's tables live: the caller's workspace when there is one on this thread and it is large enough for every plane,
  // else the static arena (its cache is locked from the first lookup to the launch: builds and the kernel are queued under the lock)
  vpf_workspace* const wsp = tables ? t_lzm_workspace : nullptr;
  LzmWorkspace* const wrec = wsp && wsp->ptr && ((uintptr_t)wsp->ptr & 255u) == 0 ? reinterpret_cast<LzmWorkspace*>(wsp->opaque) : nullptr;
  bool arena_locked = false;
  int arena_ids[6], n_arena = 0;
  u32x4* arena = nullptr;
  if (tables) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
    capturing = cs != hipStreamCaptureStatusNone;
  }
  uint32_t ws_touched = 0;  // the workspace entries this launch has been given (hit or added): never dropped under it
  // one table of one plane: -> its address (nullptr: evaluate the weights in the kernel), building it first where nobody has
  auto table = [&](uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes, auto&& build) -> const u32x4* {
    if (wrec) {
      const LzmTableCache::Hit h = wrec->get(wsp->bytes, st, dev, capturing, kind, k0, k1, k2, k3, bytes, &ws_touched);
      if (h.off16) {
        u32x4* const p = static_cast<u32x4*>(wsp->ptr) + h.off16;
        if (h.build) { (void)hipGetLastError(); build(p); }
        return p;
      }
    }
    if (!arena_locked) { lzm_tables().begin(dev); arena_locked = true; arena = lzm_arena_base(dev); }
    if (!arena) return nullptr;
    const LzmTableCache::Hit h = lzm_tables().get(st, dev, capturing, kind, k0, k1, k2, k3, bytes);
    if (!h.off16) return nullptr;
    if (h.build) { (void)hipGetLastError(); build(arena + h.off16); lzm_tables().built(h.id, st, capturing); }
    arena_ids[n_arena++] = h.id;
    return arena + h.off16;
  };
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    const float scx = (float)j.sw / (float)j.dw, scy = (float)j.sh / (float)j.dh;
    const uint32_t rows = band_tiles << rts;
    if (tables) {
      const uint32_t strips = (j.dw * (uint32_t)j.ch + 16u * nt - 1) / (16u * nt), bands = (j.dh + rows - 1) / rows, gpb = (rows + (4u << rts) - 1) >> (rts + 2u);
      wt.ctab[p] = table(0, (uint32_t)j.ch, j.sw, j.dw, (uint32_t)nt | (uint32_t)kc << 8, (uint64_t)strips * nt * kc * 2048u, [&](u32x4* out) {
        if (nt == 8) hipLaunchKernelGGL((k_lzm_build_cols<8, 1>), dim3(strips), dim3(64), 0, st, (uint32_t)j.ch, j.sw, j.dw, scx, out);
        else if (kc == 2) hipLaunchKernelGGL((k_lzm_build_cols<4, 2>), dim3(strips), dim3(64), 0, st, (uint32_t)j.ch, j.sw, j.dw, scx, out);
        else if (kc == 3) hipLaunchKernelGGL((k_lzm_build_cols<2, 3>), dim3(strips), dim3(64), 0, st, (uint32_t)j.ch, j.sw, j.dw, scx, out);
        else hipLaunchKernelGGL((k_lzm_build_cols<4, 1>), dim3(strips), dim3(64), 0, st, (uint32_t)j.ch, j.sw, j.dw, scx, out);
      });
      wt.rtab[p] = table(1, j.sh, j.dh, rows, rts | (uint32_t)up2 << 8, (uint64_t)bands * gpb * kLzmWmBytes, [&](u32x4* out) {
        hipLaunchKernelGGL(k_lzm_build_rows, dim3(gpb, bands), dim3(64), 0, st, j.sh, j.dh, scy, rows, rts, (uint32_t)up2, out);
      });
    }
    t.g[p] = PlaneGeom{j.sw, j.sh, j.dw, j.dh, scx, scy, 0, pitch, rows, wave_lds, rts};
    t.k[p] = (uint32_t)j.k; t.ch[p] = (uint32_t)j.ch; t.by0[p] = gy;
    const uint32_t strips_p = (j.dw * j.ch + 16u * nt - 1) / (16u * nt), bxs = pair ? (strips_p + 1) / 2 : (strips_p + 3) / 4;
    gx = bxs > gx ? bxs : gx;
    gy += (j.dh + rows - 1) / rows;
  }
  // leaves the arena's lock on every way out; the launch's entries get their "last use" event once the kernel is queued (not under capture:
  // an event recorded there would be a node of the graph, and entries a graph was captured with are simply not handed back: see below)
  struct ArenaUnlock {
    bool& locked; hipStream_t st; int dev; bool capturing; int* ids; int& n; bool launched = false;
    ~ArenaUnlock() {
      if (!locked) return;
      if (launched && n && !capturing) lzm_tables().used(st, dev, ids, n);
      lzm_tables().end();
    }
  } unlock{arena_locked, st, dev, capturing, arena_ids, n_arena};
  const dim3 grid(gx, gy, n);
  // a wave's LDS = the staged tile + the out tile, or — larger for the narrow strips — the scratch the column weights are composed in: not
  // needed when every plane's column table is there (a 4-tile up-scale strip: 48 -> 31 KiB per workgroup, a fourth workgroup per CU)
  uint32_t lds = pair ? lzm_pair_group_lds(lzm_pf_of(span)) : plan.group_lds;
  if (!pair && tables) {
    bool all = true;
    for (int p = 0; p < njobs; p++) all = all && wt.ctab[p] != nullptr;
    const uint32_t run = 16u * pitch + 16u * lzm_out_pitch(nt);
    if (all && run < wave_lds) {
      for (int p = 0; p < njobs; p++) t.g[p].a2 = run;
      lds = 4u * run + 2u * kLzmWmBytes;
    }
  }
  if (pair) for (int p = 0; p < njobs; p++) if (!wt.ctab[p] || !wt.rtab[p]) return false;  // the two-role form reads both tables
  if (lds > 64u * 1024u && !(kc == 3 ? (lzm_pf_of(span, 3) == 8 ? lzm_big_lds_ok<LzMfma2k8>() : lzm_big_lds_ok<LzMfma2k6>()) : kc == 2 ? (lzm_pf_of(span, 2) == 8 ? lzm_big_lds_ok<LzMfma4k8>() : lzm_pf_of(span, 2) == 6 ? lzm_big_lds_ok<LzMfma4k6>() : lzm_big_lds_ok<LzMfma4k4>())
                             : nt == 8 ? (span > 4u * 64u ? lzm_big_lds_ok<LzMfma8w>() : lzm_big_lds_ok<LzMfma8>()) : lzm_big_lds_ok<LzMfma4>())) return false;
#define VPF_LZM_GO(K) do { if (log_level() >= 2 || trace_on()) note_kernel("k_lanczos_mfma<" #K ">"); (void)hipGetLastError(); \
                           if (n <= (uint32_t)kSmallBatch) hipLaunchKernelGGL((k_lanczos_mfma<K, BatchArgs>), grid, dim3(256), lds, st, small_batch(a, n), t, wt); \
                           else hipLaunchKernelGGL((k_lanczos_mfma<K, BatchArgsL>), grid, dim3(256), lds, st, a, t, wt); \
                           unlock.launched = true; } while (0)
  if (kc == 3 && lzm_pf_of(span, 3) == 8) VPF_LZM_GO(LzMfma2k8);
  else if (kc == 3) VPF_LZM_GO(LzMfma2k6);
  else if (kc == 2 && lzm_pf_of(span, 2) == 8) VPF_LZM_GO(LzMfma4k8);
  else if (kc == 2 && lzm_pf_of(span, 2) == 6) VPF_LZM_GO(LzMfma4k6);
  else if (kc == 2) VPF_LZM_GO(LzMfma4k4);
#ifdef VPF_LAB_FORMS
  else if (pair && nt == 8 && narrow) VPF_LZM_GO(LzPairN);
  else if (pair && nt == 8 && span > 4u * 64u) VPF_LZM_GO(LzPairW);
  else if (pair && nt == 8) VPF_LZM_GO(LzPair);
#endif
  else if (nt == 8 && up2 && !narrow) VPF_LZM_GO(LzMfma8uw);
  else if (nt == 8 && up2) VPF_LZM_GO(LzMfma8u);
  else if (up2) VPF_LZM_GO(LzMfma4u);
  else if (nt == 8 && narrow) VPF_LZM_GO(LzMfma8n);
  else if (nt == 8 && span > 4u * 64u) VPF_LZM_GO(LzMfma8w);
  else if (nt == 8) VPF_LZM_GO(LzMfma8);
  else if (narrow) VPF_LZM_GO(LzMfma4n);
  else VPF_LZM_GO(LzMfma4);
#undef VPF_LZM_GO
  return true;
}

}  // namespace vpf
VPF_WAVE_TIMES_EXPORT(vpf_lab_wave_times_lanczos)
