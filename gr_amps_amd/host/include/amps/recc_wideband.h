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
#include <string>

namespace gr {
namespace amps {

class AMPS_API recc_wideband : virtual public gr::sync_block {
public:
    typedef AMPS_SPTR<recc_wideband> sptr;
    // slicer: -1 = the library default (spec D since round 4), 0 = numeric spec A (arctangent discriminator + boxcar), 1 = spec B,
    //         2 = spec C, 3 = spec D (include/amps_recc_numerics.h)
    // groups / group: one band over the GPUs of a node (BASELINE configs[4]).  groups = 2, 4 or 8: this block (one flow graph and one
    //         process per GPU, every one fed the same stream) decodes interleaved channel group `group` only -- cfg.wideband_groups of
    //         include/amps_recc.h; channel numbers on the ports stay whole-band numbers.
    // decim: input samples per filter-bank frame, 512 (60 ksps per channel) or 768 (40 ksps); 0 = amps_recc_default_wideband_decim()
    static sptr make(int n_channels = 832, int first_bin = 96, int slicer = -1, int groups = 0, int group = 0, int decim = 0);
    // Let ONE rank own the stream: after this call (a collective over all `nranks` blocks; `id` = the 128 bytes one of them got from
    // rccl_unique_id(), carried between the processes by the application) work() distributes rank `root`'s input over xGMI with RCCL
    // inside amps_recc_push_wideband_dist (mode 0 = flat broadcast, 1 = scatter + all-gather: AMPS_RECC_DIST_*).  The other ranks'
    // input only paces their flow graphs and is ignored -- INCLUDING ITS ITEM COUNTS: GNU Radio's schedulers in different processes do
    // not hand out the same noutput_items sequence, so the block never puts its own count into a collective.  The root cuts its input
    // into blocks of at most 2^22 samples, one collective per block, each carrying its size in the library's header; the ranks stay in
    // step by STREAM POSITION: a non-root rank joins collectives -- of whatever size the root announces -- until the root's stream has
    // covered the items its own pacing input has offered (feed it a null source behind a throttle at the root's sample rate).
    // A rank whose push or drain fails aborts its communicator before it returns WORK_DONE
    // (amps_recc_rccl_abort): its peers' bounded waits then end with -ETIMEDOUT instead of never.
    virtual void set_rccl(const std::string &id, int nranks, int rank, int root = 0, int mode = 0) = 0;
    static std::string rccl_unique_id();
};

} // namespace amps
} // namespace gr
