// K1 launch geometry shared by the Gram kernels (gram.hip: fp32 / fp64 MFMA; gram_split.hip: split-bf16 MFMA).
#pragma once

#include "hip_common.h"

namespace ccz {

// blockIdx -> (tile, row chunk), XCD-aware.  The dispatcher is observed to place block b on XCD b % 8
// (a speed hint only: any placement gives the same result).  The tile list is in supertile order
// (4 x 8 blocks of tiles); XCD x owns the contiguous slice [x * per_xcd, (x+1) * per_xcd) of it and walks
// that slice chunk after chunk, so the ~32 workgroups an XCD runs at any time are neighbouring tiles that
// share panels through that XCD's 4 MiB L2, while all XCDs stream the same row chunk (Infinity Cache).
// per_xcd == 0 selects the plain chunk-major order (small grids: one workgroup per CU matters more).
struct WorkItem { int tile; int64_t chunk; bool valid; };
__device__ __forceinline__ WorkItem locate_work(unsigned bid, int ntiles, int per_xcd, int64_t ksplit) {
  WorkItem it;
  if (per_xcd == 0) {
    it.tile = int(bid % unsigned(ntiles));
    it.chunk = bid / unsigned(ntiles);
    it.valid = it.chunk < ksplit;
    return it;
  }
  const unsigned x = bid & 7u, m = bid >> 3;
  if (per_xcd < 0) {
    // chunk-per-XCD order: XCD x works through row chunks x, x + 8, ... and walks the WHOLE tile list for each, so
    // the 32 workgroups it runs at any time are 32 consecutive tiles of one chunk -- a 4 x 8 supertile that shares 12
    // panels through that XCD's L2.  Odd local chunks walk the list backwards: the list's ragged tail (528 tiles =
    // 16.5 rounds of 32) meets the tail of the next chunk and the two half rounds fill the XCD together.
    // XCD x enters its (cyclic) sequence x * rot items in, rot a multiple of 32: at any moment the eight XCDs are on
    // different supertiles (no two flush the same tile of G or stream the same columns at the same time)
    const unsigned total = unsigned(ksplit >> 3) * unsigned(ntiles);
    const unsigned rot = (total >> 8) << 5;
    unsigned mm = m + x * rot;
    if (mm >= total) mm -= total;
    const unsigned lc = mm / unsigned(ntiles);
    unsigned t = mm - lc * unsigned(ntiles);
    if (lc & 1u) t = unsigned(ntiles) - 1u - t;
    it.tile = int(t);
    it.chunk = int64_t(lc) * 8 + x;
    it.valid = it.chunk < ksplit;
    return it;
  }
  it.chunk = m / unsigned(per_xcd);
  it.tile = int(x) * per_xcd + int(m % unsigned(per_xcd));
  it.valid = it.chunk < ksplit && it.tile < ntiles;
  return it;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const void* base, int64_t bytes) {
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(p));
  const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(p >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane(unsigned(bytes));
  void* q = reinterpret_cast<void*>((uint64_t(hi) << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, int(nb), 0x00020000);
}

}  // namespace ccz
