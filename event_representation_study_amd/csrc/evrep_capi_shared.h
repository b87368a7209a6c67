// evrep_capi_shared.h -- what the translation units of the extern "C" surface share on the host side (r05: the library is
// built from four translation units compiled in parallel -- evrep_capi.hip: plan / binning / read-backs, evrep_capi_mdes.hip:
// MixedDensityEventStack, evrep_capi_builders.hip: the other builders, evrep_capi_gwd.hip: GWD / OTMI clouds / entropic GW).
// No allocation, no global state besides the thread-local last HIP error string, no environment variable read.
#pragma once
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "evrep_common.h"

#define EVREP_HIDDEN __attribute__((visibility("hidden")))

namespace evrep_host {
EVREP_HIDDEN char *last_error_buf();       // the calling thread's 256-byte error string (evrep_capi.hip)
EVREP_HIDDEN int hip_check(hipError_t e, const char *what);
// the pixel-sorted stream + chunk offsets + WindowMeta from the runs of k_block_keysort (evrep_capi.hip)
EVREP_HIDDEN int column_sort_keys(const evrep_plan *plan, const int32_t *events, const int64_t *offsets, void *workspace, hipStream_t stream);
}  // namespace evrep_host

#define LAUNCH_CHECK(what)                                              \
    do {                                                                \
        int rc_ = evrep_host::hip_check(hipGetLastError(), what);       \
        if (rc_ != EVREP_OK) return rc_;                                \
    } while (0)

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static inline int check_common(const evrep_plan *plan, const void *events, const void *offsets, const void *ws) {
    if (!plan || plan->abi_version != EVREP_ABI_VERSION || !offsets || !ws) return EVREP_EINVAL;
    if (plan->total_events > 0 && !events) return EVREP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(events) & 15u) || (reinterpret_cast<uintptr_t>(ws) & 255u)) return EVREP_EINVAL;
    return EVREP_OK;
}

#define WS(type, field) reinterpret_cast<type *>(static_cast<char *>(workspace) + plan->field)
#define CWS(type, field) reinterpret_cast<const type *>(static_cast<const char *>(workspace) + plan->field)

constexpr double kKeySortedMaxPerUnit = 220.0;   // average records per builder unit up to which the key-sorted pass is chosen (r04: 110 -> 220, the warm path of the builders beats the per-key column sort for a single builder per binning pass up to 500 000 events on 640x480: bin + build 159 vs 163 us for ERGO-12, 91 vs 115 us for EventStack)
constexpr double kDeepStageMinPerUnit = 110.0;   // classic passes: windows denser than this stage 256 records per unit (stage_classic)
