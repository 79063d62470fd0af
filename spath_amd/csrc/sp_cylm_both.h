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
constexpr uint32_t kMBigSceneTris = 32768u;       // from here on the 512-thread shape (measured crossover between 10^4 and 10^5 triangles)
constexpr uint32_t kMIdxBits = cylm256::kMIdxBits;
}
