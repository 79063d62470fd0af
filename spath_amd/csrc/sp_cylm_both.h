// The default scan (sp_cylm_scan.h) in its two workgroup shapes: sp::cylm256 and sp::cylm512.
#pragma once

#include "sp_cyl_scan.h"
#include "sp_radix_sort.h"

#include <type_traits>

#define SP_CYLM_NS cylm256
#define SP_CYLM_THREADS 256
#ifdef SP_CYLM_GROUP_256
#define SP_CYLM_GROUP SP_CYLM_GROUP_256
#endif
#include "sp_cylm_scan.h"
#undef SP_CYLM_NS
#undef SP_CYLM_THREADS
#undef SP_CYLM_TILE
#undef SP_CYLM_GROUP

#define SP_CYLM_NS cylm512
#define SP_CYLM_THREADS 512
#ifndef SP_CYLM_GROUP_512
#define SP_CYLM_GROUP_512 8
#endif
#define SP_CYLM_GROUP SP_CYLM_GROUP_512
#include "sp_cylm_scan.h"
#undef SP_CYLM_NS
#undef SP_CYLM_THREADS
#undef SP_CYLM_TILE
#undef SP_CYLM_GROUP

namespace sp {
// from here on the 512-thread shape.  With octet bits it overtakes the 256-thread shape between 8192 and 12288 triangles of the closed room (tools/shape_crossover.py:
// 0.98 / 1.02 / 1.03 / 1.07 / 1.08 / 1.12 at 8192 / 12288 / 16384 / 24576 / 32768 / 65536); scenes whose triangles are large for their number lose with it (stage 2
// heavy: -25 % on the large-triangle scene at 10^4), hence not the very crossover.
constexpr uint32_t kMBigSceneTris = 16384u;
constexpr uint32_t kMIdxBits = cylm256::kMIdxBits;
}
