// gr::amps::recc_wideband -- NEW block type (not in the reference): the whole AMPS band in one block.
// Input: ONE gr_complex stream at 1024 x 30 kHz = 30.72 Msps (fc32); the block runs the 1024-branch polyphase channelizer with the
// RECC front end fused behind its FFT on the MI355X (amps_recc_push_wideband) and stands where the reference would need, PER
// CHANNEL, the chain freq_xlating_fir_filter_ccc -> analog_quadrature_demod_cf -> digital_clock_recovery_mm_ff ->
// digital_binary_slicer_fb -> amps_recc (grc/recctest.grc:889-937, 458, 846-874, 807, 310), 832 times over.
// Message ports:  "bursts"  pmt::cons(from_long(channel), blob(3374)) -- cdr = what amps_recc publishes (lib/recc_impl.cc:126);
//                 "records" pmt::cons(from_long(channel), blob(amps_recc_burst_t)) -- the burst already decoded.
// channel 0 = FFT bin `first_bin` (centre first_bin x 30 kHz above the stream's centre, modulo the sample rate).
#pragma once
#include <amps/api.h>

namespace gr {
namespace amps {

class AMPS_API recc_wideband : virtual public gr::sync_block {
public:
    typedef AMPS_SPTR<recc_wideband> sptr;
    // slicer: 0 = numeric spec A (discriminator + boxcar), 1 = spec B, 2 = spec C (include/amps_recc_numerics.h)
    static sptr make(int n_channels = 832, int first_bin = 96, int slicer = 0);
};

} // namespace amps
} // namespace gr
