// recc_channelizer.hip.h -- polyphase channelizer front end (placeholder until the kernel lands).
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include "amps_recc.h"

namespace amps {

struct ChannelizerState {
    bool enabled = false;
};

inline int channelizer_create(ChannelizerState &, const amps_recc_cfg_t &, hipStream_t) { return -ENOSYS; }
inline int channelizer_reset(ChannelizerState &, hipStream_t) { return 0; }
inline void channelizer_destroy(ChannelizerState &) {}
inline int channelizer_run(ChannelizerState &, const float2 *, size_t, int, hipStream_t, const float2 **, uint64_t *, uint32_t *)
{
    return -ENOSYS;
}

} // namespace amps
