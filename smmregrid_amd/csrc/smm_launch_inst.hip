// One (X dtype, Y dtype) pair of the kernel launch templates; built with -DSMM_XT=... -DSMM_YT=...
#include "smm_launch.hpp"

namespace smm_launch {
template int launch_sell<SMM_XT, SMM_YT>(const ApplyArgs&, int64_t, bool, unsigned, hipStream_t);
template int launch_tile<SMM_XT, SMM_YT>(const ApplyArgs&, int64_t, int, int64_t, int64_t, int, bool, unsigned,
                                         hipStream_t);
template int launch_sb<SMM_XT, SMM_YT>(const SbArgs&, bool, unsigned, hipStream_t);
template int launch_sb_group<SMM_XT, SMM_YT>(const SbGroupArgs&, bool, unsigned, hipStream_t);
}  // namespace smm_launch
