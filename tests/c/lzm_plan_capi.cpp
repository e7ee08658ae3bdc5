// C wrapper around vpf_lzm_plan.h for tests/test_lzm_plan_cpu.py (g++, no HIP): the launch planner, the fallback arena's table cache with a
// RECORDING stand-in for the HIP events it orders itself by, and the caller-owned workspace record
#include "vpf_lzm_plan.h"

#include <cstring>
#include <string>

namespace {
// events are small integers; "complete" when the test says so.  Every call is appended to a log the test reads back.
struct FakeSync final : vpf::LzmSync {
  std::string log;
  uint64_t next = 1, completed_upto = 0;  // events with id <= completed_upto are done
  int live = 0;
  bool alive[64];
  FakeSync() { for (bool& a : alive) a = true; }
  void* record(const void* stream, int dev) override {
    live++;
    log += "R" + std::to_string(next) + "@" + std::to_string((uint64_t)(uintptr_t)stream) + "d" + std::to_string(dev) + " ";
    return reinterpret_cast<void*>((uintptr_t)next++);
  }
  bool done(void* ev) override { return (uint64_t)(uintptr_t)ev <= completed_upto; }
  void wait(const void* stream, void* ev) override { log += "W" + std::to_string((uint64_t)(uintptr_t)ev) + "@" + std::to_string((uint64_t)(uintptr_t)stream) + " "; }
  void destroy(void*) override { live--; }
  bool device_alive(int dev) override { const bool a = alive[dev]; alive[dev] = true; return a; }
  uint64_t capturing = 0, gone = 0;  // one stream each the test declares "under capture" / "destroyed"
  int stream_state(const void* stream) override {
    const uint64_t s = (uint64_t)(uintptr_t)stream;
    return s && s == capturing ? 1 : s && s == gone ? 2 : 0;
  }
};
struct Cache {
  FakeSync sync;
  vpf::LzmTableCache cache;
  explicit Cache(uint64_t bytes) : cache(bytes, &sync) {}
};
}  // namespace

extern "C" {
// planes: njobs x {ch, sw, sh, dw, dh}; tables: bit 0 weight tables, bit 1 never the ring of two; out: {ok, nt, band_tiles, span, pitch, wave_lds, group_lds, kc, rts, up2}
void lzp_plan(int njobs, const uint32_t* planes, uint32_t n, int forced, int tables, uint32_t* out) {
  vpf::LzmPlaneIn in[3];
  for (int p = 0; p < njobs && p < 3; p++) in[p] = vpf::LzmPlaneIn{(int)planes[5 * p], planes[5 * p + 1], planes[5 * p + 2], planes[5 * p + 3], planes[5 * p + 4]};
  const vpf::LzmPlan q = vpf::lzm_plan(njobs, in, n, forced, (tables & 1) != 0, !(tables & 2));
  out[0] = q.ok; out[1] = (uint32_t)q.nt; out[2] = q.band_tiles; out[3] = q.span; out[4] = q.pitch; out[5] = q.wave_lds; out[6] = q.group_lds; out[7] = (uint32_t)q.kc; out[8] = (uint32_t)q.rts; out[9] = q.up2;
}
void* lzp_cache_new(uint64_t arena_bytes) { return new Cache(arena_bytes); }
void lzp_cache_free(void* c) { delete static_cast<Cache*>(c); }
// one launch with ONE table: begin, get, [built], used, end -> off16 | build << 31
uint32_t lzp_cache_launch(void* c, uint64_t stream, int dev, int capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes) {
  Cache* C = static_cast<Cache*>(c);
  C->cache.begin(dev);
  const vpf::LzmTableCache::Hit h = C->cache.get(reinterpret_cast<const void*>(stream), dev, capturing != 0, kind, k0, k1, k2, k3, bytes);
  if (h.off16 && h.build) C->cache.built(h.id, reinterpret_cast<const void*>(stream), capturing != 0);
  if (h.off16 && !capturing) C->cache.used(reinterpret_cast<const void*>(stream), dev, &h.id, 1);
  C->cache.end();
  return h.off16 | (h.build ? 0x80000000u : 0u);
}
// a launch with TWO tables (a plane's columns and rows): both must come out of one launch without evicting each other
void lzp_cache_launch2(void* c, uint64_t stream, int dev, const uint32_t* ka, uint64_t bytes_a, const uint32_t* kb, uint64_t bytes_b, uint32_t* out) {
  Cache* C = static_cast<Cache*>(c);
  C->cache.begin(dev);
  const void* st = reinterpret_cast<const void*>(stream);
  const vpf::LzmTableCache::Hit a = C->cache.get(st, dev, false, ka[0], ka[1], ka[2], ka[3], ka[4], bytes_a);
  if (a.off16 && a.build) C->cache.built(a.id, st, false);
  const vpf::LzmTableCache::Hit b = C->cache.get(st, dev, false, kb[0], kb[1], kb[2], kb[3], kb[4], bytes_b);
  if (b.off16 && b.build) C->cache.built(b.id, st, false);
  const int ids[4] = {a.id, b.id, b.id, a.id};  // (planes of equal shape share their tables: an entry may be named more than once)
  C->cache.used(st, dev, ids, 4);
  C->cache.end();
  out[0] = a.off16 | (a.build ? 0x80000000u : 0u); out[1] = b.off16 | (b.build ? 0x80000000u : 0u);
}
uint64_t lzp_cache_used(void* c, int dev) { Cache* C = static_cast<Cache*>(c); C->cache.begin(-1); const uint64_t u = C->cache.used_bytes(dev); C->cache.end(); return u; }
uint32_t lzp_cache_entries(void* c, int dev) { Cache* C = static_cast<Cache*>(c); C->cache.begin(-1); const uint32_t u = C->cache.entries(dev); C->cache.end(); return u; }
void lzp_sync_complete(void* c, uint64_t upto) { static_cast<Cache*>(c)->sync.completed_upto = upto; }
uint64_t lzp_sync_next(void* c) { return static_cast<Cache*>(c)->sync.next; }
int lzp_sync_live(void* c) { return static_cast<Cache*>(c)->sync.live; }
void lzp_sync_stream_states(void* c, uint64_t capturing, uint64_t gone) { static_cast<Cache*>(c)->sync.capturing = capturing; static_cast<Cache*>(c)->sync.gone = gone; }
void lzp_sync_reset_device(void* c, int dev) { static_cast<Cache*>(c)->sync.alive[dev] = false; }
// copies the log (and clears it) -> length
int lzp_sync_log(void* c, char* out, int cap) {
  Cache* C = static_cast<Cache*>(c);
  const int n = (int)C->sync.log.size() < cap - 1 ? (int)C->sync.log.size() : cap - 1;
  std::memcpy(out, C->sync.log.data(), (size_t)n);
  out[n] = 0;
  C->sync.log.clear();
  return n;
}
// the workspace record: ws = 40 zeroed uint64 the caller owns -> off16 | build << 31
uint32_t lzp_ws_get(uint64_t* opaque, uint64_t region_bytes, uint64_t stream, int dev, int capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3,
                    uint64_t bytes, uint32_t* touched) {
  vpf::LzmWorkspace* w = reinterpret_cast<vpf::LzmWorkspace*>(opaque);
  const vpf::LzmTableCache::Hit h = w->get(region_bytes, reinterpret_cast<const void*>(stream), dev, capturing != 0, kind, k0, k1, k2, k3, bytes, touched);
  return h.off16 | (h.build ? 0x80000000u : 0u);
}
uint32_t lzp_ws_n(const uint64_t* opaque) { return reinterpret_cast<const vpf::LzmWorkspace*>(opaque)->n; }
uint64_t lzp_table_bytes_bound(int ch, uint32_t dw, uint32_t dh) { return vpf::lzm_table_bytes_bound(ch, dw, dh); }
}
