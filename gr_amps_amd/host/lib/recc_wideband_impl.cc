// recc_wideband_impl.cc -- gr::amps::recc_wideband: one 30.72 Msps complex stream in, (channel, burst) and (channel, record) pairs out.
#include <amps/recc_wideband.h>
#include <cerrno>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "amps_recc.h"

namespace gr {
namespace amps {

class recc_wideband_impl : public recc_wideband {
    amps_recc_t *d_handle;
    std::vector<amps_recc_burst_t> d_recs;
    std::vector<unsigned char> d_bursts;
    static const int kMaxPush = 1 << 22;          // wideband samples per push (8192 frames)
    static const int kMaxRecs = 4096;

public:
    recc_wideband_impl(int C, int first_bin, int slicer)
        : gr::sync_block("recc_wideband", gr::io_signature::make(1, 1, 2 * sizeof(float)), gr::io_signature::make(0, 0, 0)), d_handle(nullptr),
          d_recs(kMaxRecs), d_bursts((size_t)kMaxRecs * AMPS_RECC_CAPTURE_SYMS)
    {
        amps_recc_cfg_t cfg = {};
        cfg.struct_size = sizeof(cfg);
        cfg.n_channels = (uint32_t)C;
        cfg.samples_per_symbol = 3;                // 60 ksps per channel behind the channelizer
        cfg.max_samples_per_push = kMaxPush / 512 + 72;
        cfg.max_bursts = kMaxRecs;
        cfg.device = -1;
        cfg.flags = AMPS_RECC_FLAG_KEEP_BURSTS | (slicer == 1 ? AMPS_RECC_FLAG_SLICER_PRODUCT : slicer == 2 ? AMPS_RECC_FLAG_SLICER_SINE : 0u);
        cfg.wideband_channels = 1024;
        cfg.wideband_decim = 512;
        cfg.wideband_taps_per_branch = 8;
        cfg.wideband_first_channel = (uint32_t)first_bin;
        int rc = amps_recc_create(&d_handle, &cfg);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_wideband: ") + amps_recc_strerror(rc));
        message_port_register_out(pmt::mp("bursts"));
        message_port_register_out(pmt::mp("records"));
    }
    ~recc_wideband_impl() { amps_recc_destroy(d_handle); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        const float *in = (const float *)input_items[0];
        int done = 0;
        while (done < noutput_items) {
            int n = noutput_items - done;
            if (n > kMaxPush) n = kMaxPush;
            int rc = amps_recc_push_wideband(d_handle, in + 2 * (size_t)done, (size_t)n, AMPS_MEM_HOST);
            if (rc != 0) { std::fprintf(stderr, "amps::recc_wideband: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
            size_t nrec = 0;
            rc = amps_recc_drain_bursts(d_handle, d_recs.data(), d_bursts.data(), kMaxRecs, &nrec);
            // -ENOSPC: more bursts than the list holds were found; the ones that fit are returned and the list recovers on the
            // next push -- a recoverable condition must not end the flow graph
            if (rc == -ENOSPC) std::fprintf(stderr, "amps::recc_wideband: %s (bursts dropped, continuing)\n", amps_recc_strerror(rc));
            else if (rc != 0) { std::fprintf(stderr, "amps::recc_wideband: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
            for (size_t i = 0; i < nrec; i++) {
                const pmt::pmt_t ch = pmt::from_long((long)d_recs[i].channel);
                message_port_pub(pmt::mp("bursts"), pmt::cons(ch, pmt::mp(d_bursts.data() + i * AMPS_RECC_CAPTURE_SYMS, AMPS_RECC_CAPTURE_SYMS)));
                message_port_pub(pmt::mp("records"), pmt::cons(ch, pmt::mp(&d_recs[i], sizeof(d_recs[i]))));
            }
            done += n;
        }
        consume_each(noutput_items);
        return 0;
    }
};

recc_wideband::sptr recc_wideband::make(int n_channels, int first_bin, int slicer)
{
    return gnuradio::get_initial_sptr(new recc_wideband_impl(n_channels, first_bin, slicer));
}

} // namespace amps
} // namespace gr
