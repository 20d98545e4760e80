// cpm_zone_kernels.h -- CPM_KERNEL_ZONE_LDS path (cars bucketed by zone, CDF row in LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"

namespace cpm {

struct ZoneWork {
    bool tables_dirty = true;
    void release() {}
};

template <typename F1, typename F2>
int32_t zone_resample(ZoneWork &, hipStream_t, const double *, const double *, int, int, int, int64_t, int64_t,
                      const uint32_t *, uint64_t, bool, const double *, int64_t *, int, F1, F2, std::string &err)
{
    err = "CPM_KERNEL_ZONE_LDS is not built yet";
    return CPM_ERR_STATE;
}

}  // namespace cpm
