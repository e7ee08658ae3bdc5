// C wrapper around vpf_lzm_plan.h for tests/test_lzm_plan_cpu.py (g++, no HIP)
#include "vpf_lzm_plan.h"

extern "C" {
// planes: njobs x {ch, sw, sh, dw, dh}; out: {ok, nt, band_tiles, span, pitch, wave_lds, group_lds}
void lzp_plan(int njobs, const uint32_t* planes, uint32_t n, int forced, int tables, uint32_t* out) {
  vpf::LzmPlaneIn in[3];
  for (int p = 0; p < njobs && p < 3; p++) in[p] = vpf::LzmPlaneIn{(int)planes[5 * p], planes[5 * p + 1], planes[5 * p + 2], planes[5 * p + 3], planes[5 * p + 4]};
  const vpf::LzmPlan q = vpf::lzm_plan(njobs, in, n, forced, tables != 0);
  out[0] = q.ok; out[1] = (uint32_t)q.nt; out[2] = q.band_tiles; out[3] = q.span; out[4] = q.pitch; out[5] = q.wave_lds; out[6] = q.group_lds;
}
void* lzp_cache_new(uint64_t arena_bytes) { return new vpf::LzmTableCache(arena_bytes); }
void lzp_cache_free(void* c) { delete static_cast<vpf::LzmTableCache*>(c); }
// -> off16 | build << 31
uint32_t lzp_cache_get(void* c, uint64_t stream, int dev, int capturing, uint32_t kind, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint64_t bytes) {
  const vpf::LzmTableCache::Hit h = static_cast<vpf::LzmTableCache*>(c)->get(reinterpret_cast<const void*>(stream), dev, capturing != 0, kind, k0, k1, k2, k3, bytes);
  return h.off16 | (h.build ? 0x80000000u : 0u);
}
uint64_t lzp_cache_used(void* c, int dev) { return static_cast<vpf::LzmTableCache*>(c)->used_bytes(dev); }
}
