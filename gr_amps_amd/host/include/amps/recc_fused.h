// gr::amps::recc_fused -- NEW block type (not in the reference): replaces the whole sub-chain
//   quadrature_demod_cf -> clock_recovery_mm_ff -> binary_slicer_fb -> amps_recc   (grc/ampsbs.grc:775,1752,1713,497)
// with the fused MI355X kernel.  Input: one gr_complex stream at samples_per_symbol * 20 kHz
// (200 ksps for the flow graphs as wired); output: the same "bursts" message port as gr::amps::recc,
// so amps_recc_decode connects to it unchanged.  It additionally publishes the already decoded
// record on port "records" (blob of amps_recc_burst_t).
// With xlate_rate_hz > 0 it also absorbs the flow graph's channel filter (freq_xlating_fir_filter_ccc with the
// firdes.low_pass taps, grc/recctest.grc:889-937, :115-155): the input is then the raw capture rate (400 ksps in
// recctest.grc) with the channel at xlate_center_hz, and xlate_rate_hz / xlate_decim = samples_per_symbol * 20 kHz.
// xlate_cutoff_hz / xlate_width_hz = 0 keep the flow graph's 10 kHz / 4.5 kHz.  A mobile whose carrier is off by 2 kHz loses half its
// bursts at 12 dB C/N behind THAT filter (it cuts into a signal it no longer centres) and none behind a 14 kHz one of the same length
// (profiles/r06/cfo_filter_width.txt): a receiver that expects carrier offsets sets xlate_cutoff_hz = 14e3.
#pragma once
#include <amps/api.h>

namespace gr {
namespace amps {

class AMPS_API recc_fused : virtual public gr::sync_block {
public:
    typedef AMPS_SPTR<recc_fused> sptr;
    static sptr make(int samples_per_symbol = 10, double xlate_rate_hz = 0.0, double xlate_center_hz = 0.0, int xlate_decim = 2,
                     double xlate_cutoff_hz = 0.0, double xlate_width_hz = 0.0);
};

} // namespace amps
} // namespace gr
