// posmask.h -- the positives chain of the sampled losses (embed_attribute.py:651-672, 721-745), shared by loss.hip
// and scorer.hip: user -> pos_ptr -> pos_items -> item2slot, with the optional 1-bit "in the pool?" table in front.
#pragma once
#include <stdint.h>

namespace arx {

struct PosMask {
  const int32_t* user_ids;   // [mask_rows]
  const int32_t* pos_ptr;    // CSR over users
  const int32_t* pos_items;
  const int32_t* item2slot;  // item -> column (or -1)
  // 1 bit per item: "is in the pool?" (arx_slot_map_attach_bitmap; nullable).  A user's positives are
  // random items of the catalogue and almost none of them is among the S sampled negatives: the probe
  // of the 4-byte item2slot cell (a 64-byte line per positive, from a table of 4 B x items) is
  // answered by a bit of a table 32x smaller that stays in L2 (125 KB at 1 M items).
  const uint32_t* bits;
};

__device__ __forceinline__ int pos_slot(const PosMask& pm, int item) {
  if (pm.bits && !((pm.bits[item >> 5] >> (item & 31)) & 1u)) return -1;
  return pm.item2slot[item];
}

// loss.hip: looks the attached bitmap of `item2slot` up (host side)
PosMask make_pos_mask(const int32_t* user_ids, const int32_t* pos_ptr, const int32_t* pos_items,
                      const int32_t* item2slot);

}  // namespace arx
