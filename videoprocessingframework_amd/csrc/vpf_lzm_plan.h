// vpf_lzm_plan.h — host-side planning of the matrix-core Lanczos launch (k_lanczos_mfma.hip): which strip width and band height, and the
// bookkeeping of the per-shape weight tables.  No HIP in here: the launcher includes it, and tests/test_lzm_plan_cpu.py compiles the same
// header with g++ and checks the planner against the measured sweeps in profiles/ and the table cache against its contract.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vpf_plan_bounds.h"

namespace vpf {

// ---- LDS sizes of a workgroup (four waves), shared with the kernel
constexpr uint32_t kLzmWmBytes = 4 * 2 * 64 * 16;                               // row-weight operands of four destination tiles (one "group" of 64 rows):
                                                                                 // per tile the Y operand of the ring's two K chunks (X is derived from it)
constexpr uint32_t kLzmB1Chunk = 4;                                              // N-tiles whose column-weight operands are built per pass through LDS
// staging loads per lane and source tile (PF) by strip span: 2 (rows of up to 128 B: up-scales), 4 (256 B), 5 (320 B: 2x down-scales with 8-tile
// strips), 6 / 8 (384 / 512 B: the two-chunk windows of strong down-scales); the LDS pitch of a staged row is the variant's capacity + 32 — a compile-time constant of the kernel instantiation, = 32 (mod 64)
constexpr int lzm_pf_of(uint32_t span, int kc = 1) { return kc == 3 ? (span <= 384u ? 6 : 8) : kc == 2 ? (span <= 256u ? 4 : span <= 384u ? 6 : 8) : span <= 128u ? 2 : span <= 256u ? 4 : 5; }
constexpr uint32_t lzm_pitch_of(int pf) { return 64u * (uint32_t)pf + 32u; }
constexpr uint32_t lzm_out_pitch(int nt) { return 16u * (uint32_t)nt + 16u; }   // out-transpose tile: + 16 keeps ds_write_b32 at 2-way (free)
constexpr uint32_t lzm_wave_lds(int nt, uint32_t pitch) {                         // bytes of wave-private LDS: staged tile | out tile, or the setup scratch
  const uint32_t run = 16u * pitch + 16u * lzm_out_pitch(nt), setup = 2u * kLzmB1Chunk * 1024u;
  return run > setup ? run : setup;
}
constexpr uint32_t lzm_group_lds(int nt, uint32_t pitch) { return 4u * lzm_wave_lds(nt, pitch) + 2u * kLzmWmBytes; }  // + the workgroup's two row-weight buffers
// the two-role form (LanczosPairTask): per pair one staged tile, a queue of four 4-KiB hand-over tiles, one out tile, the queue's sequence words
constexpr uint32_t lzm_pair_lds(int pf) { return 16u * lzm_pitch_of(pf) + 4u * 4096u + 16u * lzm_out_pitch(8) + 64u; }
constexpr uint32_t lzm_pair_group_lds(int pf) { return 2u * lzm_pair_lds(pf); }
constexpr uint32_t kLzmMaxLds = 80u * 1024u;  // two workgroups per CU share its 160 KB

// ---- does a plane shape fit the kernel's windows (vpf_plan_bounds.h: the tiles are walked with the kernel's own coordinate arithmetic)?
// A per-frame caller asks the same question every call: a small per-thread cache answers it.
// span4k2: 4-tile strips with 128-B windows (two K chunks in pass 1); rows_ok: 0 no | 4 a 16-row destination tile finds its source rows in the
// ring's four source tiles | 3 only a HALF tile (8 rows) does: vertical factors of ~2.9 .. 6 (the log2 of the rows a tile carries)
struct LzmShape { int ch; uint32_t sw, sh, dw, dh; uint32_t span4, span8, span4k2, span2k3; int rows_ok; bool rows_two; };  // span2k3: 2-tile strips, 192-B windows; rows_two: the ring of two holds every tile's source rows
inline LzmShape lzm_shape(int ch, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh) {
  thread_local LzmShape cache[8] = {};
  thread_local uint32_t next = 0;
  for (const LzmShape& c : cache)
    if (c.ch == ch && c.sw == sw && c.sh == sh && c.dw == dw && c.dh == dh) return c;
  const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
  LzmShape s{ch, sw, sh, dw, dh, 0, 0, 0, 0, 0, false};
  s.rows_ok = vpf_bound_lzm_rows_ok(sh, dh, scy) ? 4 : vpf_bound_lzm_rows_ok_rt(sh, dh, scy, 8u) ? 3 : 0;
  s.rows_two = s.rows_ok == 4 && sh < dh && vpf_bound_lzm_rows_two(sh, dh, scy);
  if (s.rows_ok) {
    s.span4 = vpf_bound_lzm_span(ch, sw, dw, scx, 4); s.span8 = vpf_bound_lzm_span(ch, sw, dw, scx, 8);
    if (!s.span4) s.span4k2 = vpf_bound_lzm_span_win(ch, sw, dw, scx, 4, 128u);  // (asked for only where the 64-B windows do not hold the taps)
    if (!s.span4 && !s.span4k2) s.span2k3 = vpf_bound_lzm_span_win(ch, sw, dw, scx, 2, 192u);
  }
  cache[next++ & 7] = s;
  return s;
}

// ---- launch shape = (N-tiles per wave, 16-row destination tiles per band), the same for every plane of the launch.  A staged row is at
// most PF x 4 lanes x 16 B and the workgroup's LDS must leave room for two (8-tile strips) or three (4-tile strips) workgroups per CU;
// among the shapes that fit, the cheapest by a small cost model fitted to sweeps over both at 32 / 8 / 1 frames per dispatch
// (profiles/r03_lanczos_shape_sweep_n*.txt, tools/lanczos_shape_sweep.py):
//   a wave costs S + R w (its fixed part — operand loads, first fetch, the latency chain of a short band — plus R tiles of work, w scaled
//   by the vertical factor), the launch W = sum over planes of strips-of-four x bands x frames workgroups against the resident ones
//   (512 / 768): every round, the last partial one too, costs one wave time.
// S = 2.0; w = 1.0 tile units for an 8-tile strip, 0.45 / 0.9 / 0.8 for a 4-tile strip of a 1- / 2- / 3-channel plane (a 4-tile strip of a
// 3-channel plane is 21 pixels under the same 64-B windows; the interleaved chroma plane of NV12 wants the wide strips), scaled by the
// vertical factor as 0.5 + 0.5 scy / 1.5 (8 tiles) or 0.3 + 0.7 scy / 1.5 (4 tiles: narrow strips pay more for the extra source rows);
// bands are at least two tiles high.  Refitted at the end of round 3 to sweeps that take the minimum of three interleaved passes per
// shape (clock drift over one pass had been several percent — as large as the differences being fitted): over the 27 cases the model's
// pick is within 1.6 % of the best measured shape on average (worst 8 %; tests/test_lzm_plan_cpu.py asserts <= 6 % / <= 20 % against
// the files in profiles/).  Without weight tables S is three times that.
struct LzmPlaneIn { int ch; uint32_t sw, sh, dw, dh; };
struct LzmPlan {
  bool ok;
  int nt;               // N-tiles per wave: 8 or 4
  uint32_t band_tiles;  // 16-row destination tiles per band
  uint32_t span, pitch, wave_lds, group_lds;
  int kc;               // 64-B K chunks per pass-1 window: 1; 2 (4-tile strips only) where a tile's taps spread over up to 128 source bytes —
                        // horizontal factors of ~2.2 .. 6 (1080p -> 416 x 416 in front of a network); 3 (2-tile strips) up to 192 bytes: factors up to
                        // ~10 (1080p -> 224 x 224).  Each taken only when nothing cheaper fits
  int rts;              // log2 of the destination rows a 16-row tile carries: 4, or 3 (half tiles) where some plane's vertical factor needs it
  bool up2;             // the ring of two (LanczosMfmaTask<.., UP2>): up-scales whose strips are narrow (<= 128 staged bytes per row) and whose
                        // destination tiles all lie within two source tiles — one K chunk in pass 2, three (8-tile strips) or four workgroups per CU
};
// forced: 0 policy | (nt << 8 | band tiles): measurement and test knob (either part may be 0 = policy)
inline LzmPlan lzm_plan(int njobs, const LzmPlaneIn* jobs, uint32_t n, int forced, bool tables, bool allow_up2 = true) {
  LzmPlan P{false, 0, 0, 0, 0, 0, 0, 1, 4, false};
  int rts = 4;  // one tile height for the launch: the smallest any plane needs
  for (int p = 0; p < njobs; p++) {
    const int ok = lzm_shape(jobs[p].ch, jobs[p].sw, jobs[p].sh, jobs[p].dw, jobs[p].dh).rows_ok;
    if (!ok) return P;
    rts = std::min(rts, ok);
  }
  const uint32_t rt = 1u << rts;
  auto fits = [&](int nt, int kc, LzmPlan& q) {
    uint32_t span = 0;
    for (int p = 0; p < njobs; p++) {
      const LzmShape s = lzm_shape(jobs[p].ch, jobs[p].sw, jobs[p].sh, jobs[p].dw, jobs[p].dh);
      if (!s.rows_ok) return false;
      // (a plane whose taps fit 64-B windows fits 128-B ones: its strips are 64 B longer then; the 2-tile strips of the three-chunk form are
      // walked for every plane of such a launch)
      const uint32_t sp = kc == 3 ? (s.span2k3 ? s.span2k3 : vpf_bound_lzm_span_win(jobs[p].ch, jobs[p].sw, jobs[p].dw, (float)jobs[p].sw / (float)jobs[p].dw, 2, 192u))
                          : kc == 2 ? (s.span4 ? s.span4 + 64u : s.span4k2) : nt == 8 ? s.span8 : s.span4;
      if (!sp) return false;  // some tile's taps do not fit the window
      span = std::max(span, sp);
    }
    q.span = span;
    q.pitch = lzm_pitch_of(lzm_pf_of(span, kc));
    q.wave_lds = lzm_wave_lds(nt, q.pitch);
    q.group_lds = lzm_group_lds(nt, q.pitch);
    return span <= (kc >= 2 ? 8u : nt == 8 ? 5u : 4u) * 64u && q.group_lds <= kLzmMaxLds;  // PF staging loads of 16 B per lane and row
  };
  double best = 0.0;
  for (int ci = 0; ci < 4; ci++) {
    const int cand = ci == 0 ? 8 : ci == 3 ? 2 : 4, kc = ci == 2 ? 2 : ci == 3 ? 3 : 1;
    if (kc >= 2 && P.ok) break;  // multi-chunk windows: the fallbacks, in the order of their cost
    if (kc == 3 && !(forced > 1)) {  // 2-tile strips are 32 destination BYTES: measured against the tile kernel they win on 1- and 2-channel planes
      bool packed3 = false;          // (Y 4K -> 416 x 416 1.94 -> 1.44 us, NV12 1080p -> 224 x 224 0.85 -> 0.70) and lose on packed RGB (1.30 -> 1.58):
      for (int p = 0; p < njobs; p++) packed3 = packed3 || jobs[p].ch == 3;  // profiles/r04_lanczos_three_chunk_sweep.txt.  A forced shape takes them anyway
      if (packed3) continue;
    }
    if (forced > 1 && (forced >> 8) != 0 && (forced >> 8) != cand && !(kc == 3 && (forced >> 8) == 4)) continue;  // (forced 4-tile strips that do not fit: the 2-tile form may still take the launch)
    LzmPlan q{false, cand, 0, 0, 0, 0, 0, kc, rts, false};
    if (!fits(cand, kc, q)) continue;
    // The ring of two (round 5): a candidate whose strips are narrow on an up-scale runs a different kernel — fewer instructions per tile
    // (w 0.9 of an 8-tile strip's; 4-tile strips 0.75 / 0.5 / 0.7 by channel count, beyond 32 frames 0.7 and 0.5 / 0.7 / 0.8) and more
    // resident workgroups (768, and 1024 for 4-tile strips whose tables exist: the LDS they are evaluated in is not reserved then).
    // At 1.5 x an 8-tile strip stages up to 256 B per row: the WIDE form (LzMfma8uw: two workgroups per CU, 512 slots, w 0.7 / 0.6 / 0.45 by
    // channel count, 0.7 / 0.6 / 0.5 beyond 32 frames) pays on large launches of packed RGB (1440p -> 4K x 32: 9.0 us against 10.6 with
    // 4-tile strips) and loses on small ones (720p -> 1080p: 2.46 against 2.33): it is a candidate only when the launch's volume — strip
    // groups x tiles x frames over the 512 slots — is at least 150 (1080p -> 1620p x 32, volume 108, still prefers 4-tile strips: 5.07
    // against 5.74 us; profiles/r05_lanczos_wide_ring_of_two_ab_n32.txt).  Fitted to profiles/r05_lanczos_shape_sweep_up_n*.txt
    // (tools/lab/fit_lzm_up2.py: 92 cases, mean regret 2.4 % up to 32 frames, 2.8 % beyond; worst 13 % / 10 %)
    q.up2 = allow_up2 && kc == 1 && rts == 4 && (q.span <= 128u || (cand == 8 && q.span <= 256u));
    for (int p = 0; p < njobs && q.up2; p++) q.up2 = lzm_shape(jobs[p].ch, jobs[p].sw, jobs[p].sh, jobs[p].dw, jobs[p].dh).rows_two;
    const bool wide = q.up2 && q.span > 128u;
    if (wide) {
      double vol = 0.0;
      for (int p = 0; p < njobs; p++) vol += (double)(((jobs[p].dw * jobs[p].ch + 127u) / 128u + 3u) / 4u) * (double)((jobs[p].dh + 15u) / 16u);
      if (vol * n / 512.0 < 150.0) q.up2 = false;  // (the candidate stays, as the ring-of-four 8-tile strip it was)
    }
    const double S = 2.0 * (tables ? 1.0 : 3.0), slots = q.up2 ? (q.span > 128u ? 512.0 : cand == 8 || !tables ? 768.0 : 1024.0) : cand == 8 || kc >= 2 ? 512.0 : 768.0;
    const bool widen = q.up2 && q.span > 128u;
    uint32_t tmax = 0;
    for (int p = 0; p < njobs; p++) tmax = std::max(tmax, (jobs[p].dh + rt - 1) / rt);
    const bool free_r = !(forced > 1 && (forced & 0xff));
    for (uint32_t r = free_r ? std::min(2u, tmax) : 1u; r <= std::min(tmax, 64u); r++) {
      if (forced > 1 && (forced & 0xff) && (uint32_t)(forced & 0xff) != r && !((uint32_t)(forced & 0xff) > tmax && r == std::min(tmax, 64u))) continue;
      uint64_t wgs = 0;
      double work = 0.0;
      for (int p = 0; p < njobs; p++) {
        const uint32_t tiles = (jobs[p].dh + rt - 1) / rt, gxp = ((jobs[p].dw * jobs[p].ch + 16u * cand - 1) / (16u * cand) + 3) / 4;
        wgs += (uint64_t)gxp * ((tiles + r - 1) / r) * n;
        const double scy = (double)jobs[p].sh / (double)jobs[p].dh;
        const double w = widen ? (jobs[p].ch == 3 ? 0.45 : jobs[p].ch == 2 ? 0.6 : 0.7) : q.up2 ? (cand == 8 ? 0.9 : jobs[p].ch == 3 ? 0.7 : jobs[p].ch == 2 ? 0.5 : 0.75)
                                 : cand == 8 || kc >= 2 ? 1.0 : jobs[p].ch == 3 ? 0.8 : jobs[p].ch == 2 ? 0.9 : 0.45;
        const double vert = cand == 8 ? 0.5 + 0.5 * scy / 1.5 : 0.3 + 0.7 * scy / 1.5;
        work = std::max(work, (double)std::min(r, tiles) * w * vert);
      }
      // Round 5 (up to 128 frames per dispatch).  Up to 32 frames the model above stands as fitted (every round, the last partial one
      // too, costs one wave time).  Beyond, a launch is many rounds long and the rounds blur — workgroups start as others end —; what is
      // left of the quantisation is the TAIL, one to two more wave lives at a partly empty chip, which favours shorter bands on small
      // planes (Y 1080p -> 720p: bands of 9-15 tiles, not the whole column) and charges the 4-tile strips of 1-channel planes 0.8 of an
      // 8-tile strip instead of 0.45: cost = (4 + work') (max(1, wgs / slots) + 1.5), fitted to the sweeps at 64 and 128 frames
      // (profiles/r05_lanczos_shape_sweep_n64.txt, _n128.txt: mean regret 4 % / 3 %, worst 12 % / 11 %; the round-3 model, asked about
      // 64 / 128 frames, is 9 % / 9 % off on average and 46 % / 27 % at worst: whole-column 4-tile strips on NV12 and Y 1080p -> 720p).
      double cost = (S + work) * std::ceil((double)wgs / slots);
      if (n > 32u) {
        double work2 = 0.0;
        for (int p = 0; p < njobs; p++) {
          const uint32_t tiles = (jobs[p].dh + rt - 1) / rt;
          const double scy = (double)jobs[p].sh / (double)jobs[p].dh;
          const double w = widen ? (jobs[p].ch == 3 ? 0.5 : jobs[p].ch == 2 ? 0.6 : 0.7) : q.up2 ? (cand == 8 ? 0.7 : jobs[p].ch == 3 ? 0.8 : jobs[p].ch == 2 ? 0.7 : 0.5)
                                   : cand == 8 || kc >= 2 ? 1.0 : jobs[p].ch == 3 ? 1.0 : jobs[p].ch == 2 ? 0.9 : 0.8;
          const double vert = cand == 8 ? 0.5 + 0.5 * scy / 1.5 : 0.3 + 0.7 * scy / 1.5;
          work2 = std::max(work2, (double)std::min(r, tiles) * w * vert);
        }
        cost = (4.0 * (tables ? 1.0 : 3.0) + work2) * (std::max(1.0, (double)wgs / slots) + 1.5);
      }
      if (!P.ok || cost < best) { best = cost; P = q; P.ok = true; P.band_tiles = r; }
    }
  }
  // Round 6: ONE up-scaled frame per dispatch (what an unmodified PySurfaceResizer.Execute() issues).  Such a launch is a single round of waves
  // that all start together, and its time is the time of the fullest CU: L = ceil(workgroups / 256) of them share each CU's SIMDs, a lone wave runs
  // a tile at its latency (tl), L co-resident ones at L x their issue time (ti) — so the best band height sits just under a multiple of 256
  // workgroups (RGB 1080p -> 4K: bands of 8 tiles = 765 workgroups 12.7 us, of 7 = 900 13.4, of 6 = 1 035, a second round, 16.0), which the model
  // above, linear in the band height within a round, cannot see.  The strip width stays the model's pick; the band height is re-chosen by
  //   cost = rounds x (a0 + a1 (L - 1) + work x max(tl, ti L)),
  // four constants per kernel class fitted to the n = 1 sweeps (profiles/r05_lanczos_shape_sweep_up_n1.txt, 20 cases, bands of up to 16 tiles;
  // tools/lab/fit_lzm_lone.py).  Measured at every band height from 3 to 9 tiles on one box (profiles/r06_m_lone_upscales.txt: the old pick against
  // the new one): RGB 1080p -> 4K 13.24 -> 12.41 us, RGB 1440p -> 4K 18.2 -> 16.4, NV12 1440p -> 4K 9.46 -> 8.79, NV12 1080p -> 4K level.  Where it
  // applies: the ring-of-two kernels (on the down-scales the same fit has no signal: the 4-tile strips' band heights tie in it, what it gains on
  // 4K -> 1440p it loses on 1080p -> 900p) and launches whose planes all have the same number of strip groups (packed RGB, Y, NV12) — "the
  // fullest CU holds ceil(workgroups / 256)" is wrong where they differ: YUV420 1080p -> 4K with bands of 4 tiles is 782 workgroups, 14 more than
  // three per CU, and runs 9.9 us where the 629 of 5-tile bands take 10.9 (first form of this rule, same file).  Everything else keeps the pick above.
  bool lone_rule = P.ok && n == 1u && tables && P.up2 && P.span <= 128u && !(forced > 1 && (forced & 0xff));
  for (int p = 1; p < njobs && lone_rule; p++)
    lone_rule = (jobs[p].dw * jobs[p].ch + 16u * P.nt - 1) / (16u * P.nt) == (jobs[0].dw * jobs[0].ch + 16u * P.nt - 1) / (16u * P.nt);
  if (lone_rule) {
    static const double K[2][4] = {{4.93, 0.0, 1.52, 0.86}, {4.65, 0.14, 1.42, 0.58}};  // a0 a1 tl ti: 8-tile strips, 4-tile strips
    const double* k = K[P.nt == 8 ? 0 : 1];
    const double slots = P.nt == 8 ? 768.0 : 1024.0, cap = slots / 256.0;
    uint32_t tmax = 0;
    for (int p = 0; p < njobs; p++) tmax = std::max(tmax, (jobs[p].dh + 15u) / 16u);
    double best1 = 0.0;
    uint32_t r1 = 0;
    for (uint32_t r = std::min(2u, tmax); r <= std::min(tmax, 64u); r++) {
      uint64_t wgs = 0;
      double work = 0.0;
      for (int p = 0; p < njobs; p++) {
        const uint32_t tiles = (jobs[p].dh + 15u) / 16u, gxp = ((jobs[p].dw * jobs[p].ch + 16u * P.nt - 1) / (16u * P.nt) + 3) / 4;
        wgs += (uint64_t)gxp * ((tiles + r - 1) / r);
        const double scy = (double)jobs[p].sh / (double)jobs[p].dh;
        const double w = P.nt == 8 ? 0.9 : jobs[p].ch == 3 ? 0.7 : jobs[p].ch == 2 ? 0.5 : 0.75;
        const double vert = P.nt == 8 ? 0.5 + 0.5 * scy / 1.5 : 0.3 + 0.7 * scy / 1.5;
        work = std::max(work, (double)std::min(r, tiles) * w * vert);
      }
      const double L = std::min(cap, std::ceil((double)wgs / 256.0));
      const double cost = std::ceil((double)wgs / slots) * (k[0] + k[1] * (L - 1.0) + work * std::max(k[2], k[3] * L));
      if (!r1 || cost < best1) { best1 = cost; r1 = r; }
    }
    if (r1) P.band_tiles = r1;
  }
  // The same question for ONE down-scaled frame per dispatch (the ring of four).  Here a band's ring fill counts — a band of r destination tiles walks
  // r scy + 2 source tiles, three per destination tile at r = 2 and 2.5 at r = 4 — and both strip widths are candidates again:
  //   cost = rounds x (a0 + work x max(1, ti L)),   work = max over planes of w_ch (cs (r scy + 2) + r),   L = min(cap, ceil(workgroups / 256)),
  // fitted per strip width to TWO boxes' sweeps of every band height (profiles/r05_lanczos_shape_sweep_down_n1.txt, 16 cases, and
  // profiles/r06_p_lone_downscales.txt, 8 cases; tools/lab/fit_lzm_lone.py) and cross-validated between them: fitted on either, the other's mean regret
  // falls (2.8 % -> 1.1 % with worst 18 % -> 6 %; 7.7 % -> 3.1 % with worst 21 % -> 12 %); the decisions do not move with the constants.  RGB 4K -> 1080p
  // 15.2 -> 14.2 us (4-tile strips in bands of four), NV12 / YUV420 4K -> 1440p 10-13 % (bands of three).  Only for launches that fill the chip: where
  // the pick above has at most 1.5 workgroups per CU, fewer and longer waves have nothing to collect (YUV420 1080p -> 900p, 323 workgroups: 6.9 us as
  // picked, 7.6 with bands of three).  One-chunk windows, whole tiles, table launches, no forced shape; everything else keeps the pick above.
  bool lone_down = P.ok && n == 1u && tables && !P.up2 && P.kc == 1 && P.rts == 4 && forced <= 1;
  for (int p = 0; p < njobs && lone_down; p++) lone_down = jobs[p].sh >= jobs[p].dh;  // (no plane could take the ring of two under another strip width)
  if (lone_down) {
    static const double K[2][5] = {{4.26, 0.81, 0.91, 0.92, 0.49}, {3.32, 0.77, 1.05, 1.0, 0.15}};  // a0 ti w1 w2 cs: 8-tile strips, 4-tile strips (w3 = 1, tl = 1)
    auto shape = [&](int nt, uint32_t r, uint64_t& wgs, double& work) {
      const double* k = K[nt == 8 ? 0 : 1];
      wgs = 0;
      work = 0.0;
      for (int p = 0; p < njobs; p++) {
        const uint32_t tiles = (jobs[p].dh + 15u) / 16u, gxp = ((jobs[p].dw * jobs[p].ch + 16u * nt - 1) / (16u * nt) + 3) / 4, rr = std::min(r, tiles);
        wgs += (uint64_t)gxp * ((tiles + r - 1) / r);
        const double scy = (double)jobs[p].sh / (double)jobs[p].dh, w = jobs[p].ch == 1 ? k[2] : jobs[p].ch == 2 ? k[3] : 1.0;
        work = std::max(work, w * (k[4] * ((double)rr * scy + 2.0) + (double)rr));
      }
    };
    uint64_t wgs0;
    double work0;
    shape(P.nt, P.band_tiles, wgs0, work0);
    if (wgs0 > 384u) {
      uint32_t tmax = 0;
      for (int p = 0; p < njobs; p++) tmax = std::max(tmax, (jobs[p].dh + 15u) / 16u);
      double best1 = 0.0;
      LzmPlan B = P;
      bool have = false;
      for (int nt = 8; nt >= 4; nt -= 4) {
        LzmPlan q{false, nt, 0, 0, 0, 0, 0, 1, 4, false};
        if (!fits(nt, 1, q)) continue;
        const double* k = K[nt == 8 ? 0 : 1];
        const double slots = nt == 8 ? 512.0 : 768.0, cap = slots / 256.0;
        for (uint32_t r = std::min(2u, tmax); r <= std::min(tmax, 64u); r++) {
          uint64_t wgs;
          double work;
          shape(nt, r, wgs, work);
          const double L = std::min(cap, std::ceil((double)wgs / 256.0));
          const double cost = std::ceil((double)wgs / slots) * (k[0] + work * std::max(1.0, k[1] * L));
          if (!have || cost < best1) { best1 = cost; B = q; B.ok = true; B.band_tiles = r; have = true; }
        }
      }
      if (have) P = B;
    }
  }
  return P;
}

// ---- weight-table bookkeeping.  The operand images of a plane shape (column weights per strip, row weights per band) are built once by two
// small kernels and read by every launch of that shape.  They live
//   (a) in a WORKSPACE the caller owns (vpf_resize_workspace_bytes / vpf_resize_ws, include/vpf_hip.h: NPP's scratch-buffer pattern; the
//       Task layer's ResizeSurface allocates one per source shape next to its destination surface) — LzmWorkspace below: a handful of
//       entries, everything ordered by the ONE stream the workspace is used on, no events;
//   (b) else in a small static arena per device (4 MiB, a fallback) — LzmTableCache: hash lookup, least-recently-used eviction.  What
//       makes it safe: an entry's build is followed by an event; a launch on any stream waits for that event until it has been seen
//       complete once (then the table is final for everybody — until evicted); an entry remembers the (up to four) streams it has been used
//       on — no HIP call per launch —, and whoever evicts it records an event on each of them THEN and makes its own stream wait for those
//       before the build kernel that overwrites the space (an entry used on more streams than that gets an event per launch).  A
//       capturing stream always queues its own build (a captured build has not run; it rewrites the same bytes).  A device that was
//       reset loses its tables: the canary check flushes the host's picture of it.
// Offsets are in 16-B units from the region's base; 0 = "no table: evaluate the weights in the kernel".
struct LzmKey {
  int dev;
  uint32_t kind, k0, k1, k2, k3;  // kind 0: columns (ch, sw, dw, nt | chunks << 8) | 1: rows (sh, dh, band rows, log2 tile rows | ring of two << 8)
  bool operator==(const LzmKey& o) const { return dev == o.dev && kind == o.kind && k0 == o.k0 && k1 == o.k1 && k2 == o.k2 && k3 == o.k3; }
};
struct LzmKeyHash {
  size_t operator()(const LzmKey& k) const {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)(uint32_t)k.dev;
    for (uint32_t v : {k.kind, k.k0, k.k1, k.k2, k.k3}) h = (h ^ v) * 0xff51afd7ed558ccdull + (h >> 29);
    return (size_t)h;
  }
};
// the ordering primitives the cache needs (HIP events in the launcher, a recording fake in tests/test_lzm_plan_cpu.py)
struct LzmSync {
  virtual ~LzmSync() {}
  virtual void* record(const void* stream, int dev) = 0;  // a new event recorded on `stream` (nullptr: could not)
  virtual bool done(void* ev) = 0;                         // has it completed?  (non-blocking)
  virtual void wait(const void* stream, void* ev) = 0;     // `stream` waits for it
  virtual void destroy(void* ev) = 0;
  virtual bool device_alive(int dev) = 0;                  // false once after the device was reset: the cache forgets the device's tables
  // a stream an entry remembers from an EARLIER launch of somebody else, asked about at eviction: 0 = an event can be recorded on it |
  // 1 = it is under capture (a record would become a node of that graph and the cross-stream wait could invalidate the capture: the entry
  // is left alone) | 2 = the handle no longer names a stream (destroyed: nothing left to order behind)
  virtual int stream_state(const void* stream) { (void)stream; return 0; }
};
class LzmTableCache {
 public:
  LzmTableCache(uint64_t arena_bytes, LzmSync* sync) : cap16_((uint32_t)std::min<uint64_t>(arena_bytes / 16, 0x7fffffffu)), sync_(sync) {}
  ~LzmTableCache() { for (auto& e : ents_) drop_events(e); }
  struct Hit { uint32_t off16; bool build; int id; };  // id: handle for built() / used(); -1 with off16 == 0
  // One launch = begin() .. get() x (2 per plane) .. [built() per build queued] .. used().  The mutex is held from begin() to end(): the
  // build and the launch are QUEUED under it, so two host threads on one stream cannot interleave "entry exists" with "build queued".
  void begin(int dev) { mu_.lock(); tick_++; if (dev >= 0 && dev < 64 && !sync_->device_alive(dev)) flush(dev); }
  void end() { mu_.unlock(); }
  Hit get(const void* stream, int dev, bool capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes) {
    if (dev < 0 || dev >= 64 || !cap16_) return Hit{0, false, -1};
    const LzmKey key{dev, kind, k0, k1, k2, k3};
    auto it = map_.find(key);
    if (it != map_.end()) {
      Ent& e = ents_[it->second];
      e.tick = tick_;
      if (capturing) { e.pinned = true; return Hit{e.off16, true, it->second}; }  // a graph captured with this table replays whenever it likes: the entry stays
      if (!e.settled) {
        if (!e.built) return Hit{e.off16, true, it->second};  // its build was never followed by an event (event creation failed): build again
        if (sync_->done(e.built)) e.settled = true;
        else if (e.built_stream != stream) sync_->wait(stream, e.built);
      }
      return Hit{e.off16, false, it->second};
    }
    const uint32_t need16 = (uint32_t)std::min<uint64_t>((bytes + 255) / 256 * 16, 0xffffffffu);
    if (need16 > cap16_ - 16) return Hit{0, false, -1};
    uint32_t off = 0;
    while (!(off = first_fit(dev, need16))) {
      const int victim = lru(dev);
      if (victim < 0) return Hit{0, false, -1};  // everything left is pinned by this very launch
      Ent& v = ents_[victim];
      // the space is rewritten by a build kernel queued on `stream`: behind everything queued so far on the streams that read the old
      // table (an event recorded on each of them now), and behind its build.  The remembered handles are somebody else's and may have
      // changed since: a stream under capture keeps its entry (not a victim for this launch), a destroyed one drops out
      bool busy = false;
      for (int i = 0; i < v.nusers; i++) {
        const int state = v.users[i] == stream ? 0 : sync_->stream_state(v.users[i]);
        if (state == 1) busy = true;
        if (state == 2) { v.users[i--] = v.users[--v.nusers]; v.users[v.nusers] = nullptr; }
      }
      if (busy) { v.tick = tick_; continue; }
      for (int i = 0; i < v.nusers; i++) {
        if (v.users[i] == stream) continue;  // this stream's own earlier launches are in front of the build anyway
        void* ev = sync_->record(v.users[i], dev);
        if (ev) { sync_->wait(stream, ev); sync_->destroy(ev); }
      }
      if (v.last_use && v.last_use->ev && !sync_->done(v.last_use->ev)) sync_->wait(stream, v.last_use->ev);
      if (v.built && !v.settled && !sync_->done(v.built)) sync_->wait(stream, v.built);
      erase(victim);
    }
    int id;
    if (!free_.empty()) { id = free_.back(); free_.pop_back(); } else { id = (int)ents_.size(); ents_.emplace_back(); }
    Ent& e = ents_[id];
    e = Ent{};
    e.key = key; e.off16 = off; e.len16 = need16; e.tick = tick_; e.live = true; e.pinned = capturing;
    map_[key] = id;
    return Hit{off, true, id};
  }
  // the entry's build kernel has been queued on `stream` (not under capture: a captured build proves nothing about the arena's bytes)
  void built(int id, const void* stream, bool capturing) {
    if (id < 0 || capturing) return;
    Ent& e = ents_[id];
    if (e.built) { sync_->destroy(e.built); e.built = nullptr; }
    e.built = sync_->record(stream, e.key.dev);
    e.built_stream = stream;
    e.settled = false;
  }
  // the launch that read these entries has been queued on `stream`: the entries remember the stream (no HIP call); an entry that has met more
  // streams than it can remember gets ONE event per launch instead (shared by such entries of the launch)
  void used(const void* stream, int dev, const int* ids, int n) {
    EvRef* r = nullptr;
    for (int i = 0; i < n; i++) {
      if (ids[i] < 0) continue;
      Ent& e = ents_[ids[i]];
      bool known = false;
      for (int k = 0; k < e.nusers; k++) known = known || e.users[k] == stream;
      if (known && !e.many) continue;
      if (!known && e.nusers < 4 && !e.many) { e.users[e.nusers++] = stream; continue; }
      e.many = true;
      if (!r) r = new EvRef{sync_->record(stream, dev), 0};
      if (e.last_use == r) continue;  // the same entry twice in one launch (planes of equal shape share their tables)
      unref(e.last_use);
      e.last_use = r; r->refs++;
    }
    if (r && !r->refs) { if (r->ev) sync_->destroy(r->ev); delete r; }
  }
  uint64_t used_bytes(int dev) {
    uint64_t u = 0;
    for (const Ent& e : ents_) if (e.live && e.key.dev == dev) u += (uint64_t)e.len16 * 16;
    return u;
  }
  uint32_t entries(int dev) { uint32_t n = 0; for (const Ent& e : ents_) n += e.live && e.key.dev == dev; return n; }

 private:
  struct EvRef { void* ev; int refs; };
  struct Ent {
    LzmKey key{};
    uint32_t off16 = 0, len16 = 0;
    uint64_t tick = 0;
    void* built = nullptr;
    const void* built_stream = nullptr;
    bool settled = false, live = false, pinned = false, many = false;
    const void* users[4] = {nullptr, nullptr, nullptr, nullptr};  // streams whose launches read the table
    int nusers = 0;
    EvRef* last_use = nullptr;
  };
  void unref(EvRef*& r) {
    if (r && --r->refs == 0) { if (r->ev) sync_->destroy(r->ev); delete r; }
    r = nullptr;
  }
  void drop_events(Ent& e) {
    if (e.built) { sync_->destroy(e.built); e.built = nullptr; }
    unref(e.last_use);
  }
  void erase(int id) {
    Ent& e = ents_[id];
    drop_events(e);
    map_.erase(e.key);
    e.live = false;
    free_.push_back(id);
  }
  void flush(int dev) {
    for (int i = 0; i < (int)ents_.size(); i++)
      if (ents_[i].live && ents_[i].key.dev == dev) erase(i);
  }
  int lru(int dev) const {
    int best = -1;
    for (int i = 0; i < (int)ents_.size(); i++)
      if (ents_[i].live && !ents_[i].pinned && ents_[i].key.dev == dev && ents_[i].tick != tick_ && (best < 0 || ents_[i].tick < ents_[best].tick)) best = i;
    return best;
  }
  // lowest offset >= 16 (the first 256 B stay unused: offset 0 means "no table") where need16 units are free; 0 = nowhere
  uint32_t first_fit(int dev, uint32_t need16) const {
    std::vector<std::pair<uint32_t, uint32_t>> r;
    for (const Ent& e : ents_) if (e.live && e.key.dev == dev) r.emplace_back(e.off16, e.len16);
    std::sort(r.begin(), r.end());
    uint32_t at = 16;
    for (const auto& x : r) {
      if (x.first >= at && x.first - at >= need16) return at;
      at = std::max(at, x.first + x.second);
    }
    return (cap16_ >= at && cap16_ - at >= need16) ? at : 0;
  }
  std::mutex mu_;
  std::vector<Ent> ents_;
  std::vector<int> free_;
  std::unordered_map<LzmKey, int, LzmKeyHash> map_;
  uint64_t tick_ = 0;
  uint32_t cap16_;
  LzmSync* sync_;
};

// ---- the caller-owned workspace (vpf_workspace, include/vpf_hip.h): its `opaque` words hold this record.  Used on ONE stream at a time:
// builds and launches are ordered by that stream alone.  A launch on another stream, a captured launch or a shape the record does not
// hold rebuilds (when the space is short: everything is dropped and the region reused from its start — the new builds queue behind the
// kernels that read the old tables).  What a launch has already been GIVEN — an entry it hit as much as one it added — is never dropped
// under it: `touched` is the launch's own mask of such entries (zero at its first lookup), and a table that finds the record full while the
// mask is non-zero goes to the fallback arena instead (round 4 guarded added entries only: a hit followed by a miss on a full record handed
// out the hit table's bytes a second time — ADVICE r4).
struct LzmWorkspace {
  static constexpr uint32_t kMagic = 0x4c5a4d57u;  // "LZMW"
  static constexpr int kEntries = 8;               // three planes x {columns, rows} + two spare
  uint32_t magic, n, used16, device;
  const void* stream;
  struct E { uint32_t kind, k0, k1, k2, k3, off16, len16, pad; } e[kEntries];
  // -> off16 (0: does not fit, use the fallback) and whether the caller must queue the build
  LzmTableCache::Hit get(uint64_t region_bytes, const void* st, int dev, bool capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes,
                         uint32_t* touched) {
    const uint32_t cap16 = (uint32_t)std::min<uint64_t>(region_bytes / 16, 0x7fffffffu);
    // (a record that starts over cannot hold anything this launch was given: the launcher reads stream / device before its first lookup)
    if (magic != kMagic || device != (uint32_t)dev || n > (uint32_t)kEntries) { magic = kMagic; n = 0; used16 = 16; device = (uint32_t)dev; stream = st; *touched = 0; }
    if (stream != st) { n = 0; used16 = 16; stream = st; *touched = 0; }  // another stream: nothing in here is ordered for it — start over
    for (uint32_t i = 0; i < n; i++)
      if (e[i].kind == kind && e[i].k0 == k0 && e[i].k1 == k1 && e[i].k2 == k2 && e[i].k3 == k3) { *touched |= 1u << i; return LzmTableCache::Hit{e[i].off16, capturing, (int)i}; }
    const uint32_t need16 = (uint32_t)std::min<uint64_t>((bytes + 255) / 256 * 16, 0xffffffffu);
    if (need16 > cap16 || cap16 < 16 || need16 > cap16 - 16) return LzmTableCache::Hit{0, false, -1};
    if (n == (uint32_t)kEntries || used16 + need16 > cap16) {
      if (*touched) return LzmTableCache::Hit{0, false, -1};  // tables of this very launch would go: the fallback takes this one
      n = 0; used16 = 16;
    }
    e[n] = E{kind, k0, k1, k2, k3, used16, need16, 0};
    used16 += need16;
    *touched |= 1u << n;
    return LzmTableCache::Hit{e[n].off16, true, (int)n++};
  }
};
static_assert(sizeof(LzmWorkspace) <= 40 * 8, "fits vpf_workspace::opaque");

// upper bound of the table bytes one plane needs under ANY launch shape the planner may pick (column tables: strips x nt x K chunks x 2 KiB,
// largest with two-chunk windows; row tables: (bands x groups per band) x 8 KiB, bands of at least two tiles) — what vpf_resize_workspace_bytes adds up
inline uint64_t lzm_table_bytes_bound(int ch, uint32_t dw, uint32_t dh) {
  const uint64_t dwb = (uint64_t)dw * (uint64_t)ch;
  const uint64_t cols = ((dwb + 127) / 128 + 1) * 8 * 2048 * 3;  // (x 3: the three-chunk windows of the strongest down-scales)
  const uint64_t rows = ((uint64_t)(dh + 31) / 32 + (uint64_t)(dh + 15) / 16 + 1) * kLzmWmBytes;  // (groups of four HALF tiles: 32 rows)
  return cols + rows + 2 * 256;
}

}  // namespace vpf
