// vpf_lzm_plan.h — host-side planning of the matrix-core Lanczos launch (k_lanczos_mfma.hip): which strip width and band height, and the
// bookkeeping of the per-shape weight tables.  No HIP in here: the launcher includes it, and tests/test_lzm_plan_cpu.py compiles the same
// header with g++ and checks the planner against the measured sweeps in profiles/ and the table cache against its contract.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "vpf_plan_bounds.h"

namespace vpf {

// ---- LDS sizes of a workgroup (four waves), shared with the kernel
constexpr uint32_t kLzmWmBytes = 4 * 2 * 64 * 16;                               // row-weight operands of four destination tiles (one "group" of 64 rows):
                                                                                 // per tile the Y operand of the ring's two K chunks (X is derived from it)
constexpr uint32_t kLzmB1Chunk = 4;                                              // N-tiles whose column-weight operands are built per pass through LDS
// staging loads per lane and source tile (PF) by strip span: 2 (rows of up to 128 B: up-scales), 4 (256 B), 5 (320 B: 2x down-scales with 8-tile
// strips); the LDS pitch of a staged row is the variant's capacity + 32 — a compile-time constant of the kernel instantiation, = 32 (mod 64)
constexpr int lzm_pf_of(uint32_t span) { return span <= 128u ? 2 : span <= 256u ? 4 : 5; }
constexpr uint32_t lzm_pitch_of(int pf) { return 64u * (uint32_t)pf + 32u; }
constexpr uint32_t lzm_out_pitch(int nt) { return 16u * (uint32_t)nt + 16u; }   // out-transpose tile: + 16 keeps ds_write_b32 at 2-way (free)
constexpr uint32_t lzm_wave_lds(int nt, uint32_t pitch) {                         // bytes of wave-private LDS: staged tile | out tile, or the setup scratch
  const uint32_t run = 16u * pitch + 16u * lzm_out_pitch(nt), setup = 2u * kLzmB1Chunk * 1024u;
  return run > setup ? run : setup;
}
constexpr uint32_t lzm_group_lds(int nt, uint32_t pitch) { return 4u * lzm_wave_lds(nt, pitch) + 2u * kLzmWmBytes; }  // + the workgroup's two row-weight buffers
constexpr uint32_t kLzmMaxLds = 80u * 1024u;  // two workgroups per CU share its 160 KB

// ---- does a plane shape fit the kernel's windows (vpf_plan_bounds.h: the tiles are walked with the kernel's own coordinate arithmetic)?
// A per-frame caller asks the same question every call: a small per-thread cache answers it.
struct LzmShape { int ch; uint32_t sw, sh, dw, dh; uint32_t span4, span8; bool rows_ok; };
inline LzmShape lzm_shape(int ch, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh) {
  thread_local LzmShape cache[8] = {};
  thread_local uint32_t next = 0;
  for (const LzmShape& c : cache)
    if (c.ch == ch && c.sw == sw && c.sh == sh && c.dw == dw && c.dh == dh) return c;
  const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
  LzmShape s{ch, sw, sh, dw, dh, 0, 0, false};
  s.rows_ok = vpf_bound_lzm_rows_ok(sh, dh, scy) != 0;
  if (s.rows_ok) { s.span4 = vpf_bound_lzm_span(ch, sw, dw, scx, 4); s.span8 = vpf_bound_lzm_span(ch, sw, dw, scx, 8); }
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
};
// forced: 0 policy | (nt << 8 | band tiles): measurement and test knob (either part may be 0 = policy)
inline LzmPlan lzm_plan(int njobs, const LzmPlaneIn* jobs, uint32_t n, int forced, bool tables) {
  LzmPlan P{false, 0, 0, 0, 0, 0, 0};
  auto fits = [&](int nt, LzmPlan& q) {
    uint32_t span = 0;
    for (int p = 0; p < njobs; p++) {
      const LzmShape s = lzm_shape(jobs[p].ch, jobs[p].sw, jobs[p].sh, jobs[p].dw, jobs[p].dh);
      if (!s.rows_ok) return false;
      const uint32_t sp = nt == 8 ? s.span8 : s.span4;
      if (!sp) return false;  // some tile's taps do not fit the 64-B window
      span = std::max(span, sp);
    }
    q.span = span;
    q.pitch = lzm_pitch_of(lzm_pf_of(span));
    q.wave_lds = lzm_wave_lds(nt, q.pitch);
    q.group_lds = lzm_group_lds(nt, q.pitch);
    return span <= (nt == 8 ? 5u : 4u) * 64u && q.group_lds <= kLzmMaxLds;  // PF staging loads of 4 lanes x 16 B per row
  };
  double best = 0.0;
  for (int cand = 8; cand >= 4; cand -= 4) {
    if (forced > 1 && (forced >> 8) != 0 && (forced >> 8) != cand) continue;
    LzmPlan q{false, cand, 0, 0, 0, 0, 0};
    if (!fits(cand, q)) continue;
    const double S = 2.0 * (tables ? 1.0 : 3.0), slots = cand == 8 ? 512.0 : 768.0;
    uint32_t tmax = 0;
    for (int p = 0; p < njobs; p++) tmax = std::max(tmax, (jobs[p].dh + 15) / 16);
    const bool free_r = !(forced > 1 && (forced & 0xff));
    for (uint32_t r = free_r ? std::min(2u, tmax) : 1u; r <= std::min(tmax, 64u); r++) {
      if (forced > 1 && (forced & 0xff) && (uint32_t)(forced & 0xff) != r && !((uint32_t)(forced & 0xff) > tmax && r == std::min(tmax, 64u))) continue;
      uint64_t wgs = 0;
      double work = 0.0;
      for (int p = 0; p < njobs; p++) {
        const uint32_t tiles = (jobs[p].dh + 15) / 16, gxp = ((jobs[p].dw * jobs[p].ch + 16u * cand - 1) / (16u * cand) + 3) / 4;
        wgs += (uint64_t)gxp * ((tiles + r - 1) / r) * n;
        const double scy = (double)jobs[p].sh / (double)jobs[p].dh;
        const double w = cand == 8 ? 1.0 : jobs[p].ch == 3 ? 0.8 : jobs[p].ch == 2 ? 0.9 : 0.45;
        const double vert = cand == 8 ? 0.5 + 0.5 * scy / 1.5 : 0.3 + 0.7 * scy / 1.5;
        work = std::max(work, (double)std::min(r, tiles) * w * vert);
      }
      const double cost = (S + work) * std::ceil((double)wgs / slots);
      if (!P.ok || cost < best) { best = cost; P = q; P.ok = true; P.band_tiles = r; }
    }
  }
  return P;
}

// ---- weight-table bookkeeping.  One arena per device, bump-allocated, never freed or rewritten with other bytes.  An entry remembers the
// (up to four) streams that have queued its build: a launch on one of them is ordered behind the build by the stream itself; any other
// stream — and any stream that is being captured into a graph, whose build has not run — must queue the build again (idempotent: same
// bytes).  Offsets are in 16-B units; 0 means "no table" (the first 256 B of the arena stay unused).
struct LzmTab {
  int dev;
  uint32_t kind, k0, k1, k2, k3;  // kind 0: columns (ch, sw, dw, nt) | 1: rows (sh, dh, band rows, 0)
  uint32_t off16;
  const void* streams[4];
  int nstreams;
};
class LzmTableCache {
 public:
  explicit LzmTableCache(uint64_t arena_bytes) : cap16_(arena_bytes / 16) {}
  struct Hit { uint32_t off16; bool build; };  // off16 == 0: no room (evaluate the weights in the kernel); build: queue the build kernel on this stream
  Hit get(const void* stream, int dev, bool capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes) {
    std::lock_guard<std::mutex> lock(mu_);
    if (dev < 0 || dev >= 64) return Hit{0, false};
    LzmTab* e = nullptr;
    for (LzmTab& t : tabs_)
      if (t.dev == dev && t.kind == kind && t.k0 == k0 && t.k1 == k1 && t.k2 == k2 && t.k3 == k3) { e = &t; break; }
    if (!e) {
      uint32_t& used = used16_[dev];
      if (!used) used = 16;
      const uint64_t need16 = (bytes + 255) / 256 * 16;
      if ((uint64_t)used + need16 > cap16_) return Hit{0, false};
      tabs_.push_back(LzmTab{dev, kind, k0, k1, k2, k3, used, {}, 0});
      used += (uint32_t)need16;
      e = &tabs_.back();
    }
    bool known = false;
    for (int i = 0; i < e->nstreams; i++) known = known || e->streams[i] == stream;
    if (known && !capturing) return Hit{e->off16, false};
    if (!capturing) {
      if (e->nstreams < 4) e->streams[e->nstreams++] = stream;
      else { e->streams[0] = e->streams[1]; e->streams[1] = e->streams[2]; e->streams[2] = e->streams[3]; e->streams[3] = stream; }
    }
    return Hit{e->off16, true};
  }
  uint64_t used_bytes(int dev) { std::lock_guard<std::mutex> lock(mu_); return dev >= 0 && dev < 64 ? (uint64_t)used16_[dev] * 16 : 0; }

 private:
  std::mutex mu_;
  std::vector<LzmTab> tabs_;
  uint32_t used16_[64] = {};
  uint64_t cap16_;
};

}  // namespace vpf
