// gr::amps::recc_bank -- NEW block type (not in the reference): C instances of gr::amps::recc in ONE block.
// The reference instantiates one amps_recc per 30 kHz channel, and GNU Radio's thread-per-block scheduler calls every
// instance's work() from its own thread: 832 blocks = 832 launches + 832 synchronisations per scheduler pass.  A sync_block
// with C input streams is handed the same noutput_items on every stream, so one work() call is exactly one
// amps_recc_push_symbols on a handle with n_channels = C: one launch, one synchronise, every channel's
// lib/recc_impl.cc:93-145 state machine advanced by the same chunk (bit-exact per channel: each channel sees the chunk
// schedule the lone block would see).
// Ports: C byte inputs "in0".."in<C-1>"; message out "bursts" = pmt::cons(from_long(channel), blob(3374)) -- car = channel,
// cdr = what amps_recc publishes (lib/recc_impl.cc:126); use one amps_recc_decode per channel or demultiplex on the car.
#pragma once
#include <amps/api.h>

namespace gr {
namespace amps {

class AMPS_API recc_bank : virtual public gr::sync_block {
public:
    typedef AMPS_SPTR<recc_bank> sptr;
    static sptr make(int n_channels);
};

} // namespace amps
} // namespace gr
