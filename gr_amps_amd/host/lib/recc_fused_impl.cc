// recc_fused_impl.cc -- gr::amps::recc_fused: complex baseband in; on "bursts" the same 3374-byte symbol blob gr::amps::recc
// publishes (lib/recc_impl.cc:126), on "records" the already decoded record.
// Replaces quadrature_demod_cf -> clock_recovery_mm_ff -> binary_slicer_fb -> amps_recc with the fused
// MI355X kernel (amps_recc_push_iq / amps_recc_drain).
#include <amps/recc_fused.h>
#include <cerrno>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "amps_recc.h"

namespace gr {
namespace amps {

class recc_fused_impl : public recc_fused {
    amps_recc_t *d_handle;
    bool d_raw;
    std::vector<unsigned char> d_bursts;
    static const int kMaxPush = 1 << 20;
    static const int kMaxRecs = 64;

public:
    recc_fused_impl(int sps, double xlate_rate, double xlate_center, int xlate_decim, double xlate_cutoff, double xlate_width)
        : gr::sync_block("recc_fused", gr::io_signature::make(1, 1, 2 * sizeof(float)), gr::io_signature::make(0, 0, 0)), d_handle(nullptr),
          d_raw(xlate_rate > 0.0), d_bursts((size_t)kMaxRecs * AMPS_RECC_CAPTURE_SYMS)
    {
        amps_recc_cfg_t cfg = {};
        cfg.struct_size = sizeof(cfg);
        cfg.n_channels = 1;
        cfg.samples_per_symbol = (uint32_t)sps;
        cfg.max_samples_per_push = kMaxPush;
        cfg.max_bursts = kMaxRecs;
        cfg.device = -1;
        cfg.flags = AMPS_RECC_FLAG_KEEP_BURSTS;               // the captured symbols travel with the record
        int rc = amps_recc_create(&d_handle, &cfg);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_fused: ") + amps_recc_strerror(rc));
        if (d_raw) {
            amps_recc_xlate_cfg_t x = {};
            x.struct_size = sizeof(x);
            x.decim = (uint32_t)xlate_decim;
            x.rate_hz = xlate_rate;
            x.center_hz = xlate_center;
            x.cutoff_hz = xlate_cutoff;                    // 0 = the flow graph's 10 kHz / 4.5 kHz
            x.width_hz = xlate_width;
            rc = amps_recc_set_xlate(d_handle, &x);
            if (rc != 0) {
                amps_recc_destroy(d_handle);
                throw std::runtime_error(std::string("amps::recc_fused (xlate): ") + amps_recc_strerror(rc));
            }
        }
        message_port_register_out(pmt::mp("bursts"));          // what amps_recc publishes: connects to amps_recc_decode unchanged
        message_port_register_out(pmt::mp("records"));         // the same burst already decoded (recc_decode accepts it too)
    }
    ~recc_fused_impl() { amps_recc_destroy(d_handle); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        const float *in = (const float *)input_items[0];
        int done = 0;
        while (done < noutput_items) {
            int n = noutput_items - done;
            if (n > kMaxPush) n = kMaxPush;
            int rc = d_raw ? amps_recc_push_raw(d_handle, in + 2 * (size_t)done, (size_t)n, (size_t)n, AMPS_MEM_HOST)
                           : amps_recc_push_iq(d_handle, in + 2 * (size_t)done, (size_t)n, (size_t)n, AMPS_MEM_HOST);
            if (rc != 0) { std::fprintf(stderr, "amps::recc_fused: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
            amps_recc_burst_t recs[kMaxRecs];
            size_t nrec = 0;
            rc = amps_recc_drain_bursts(d_handle, recs, d_bursts.data(), kMaxRecs, &nrec);
            // -ENOSPC: more bursts than the list holds were found; the ones that fit are returned and the list recovers on the
            // next push -- a recoverable condition must not end the flow graph
            if (rc == -ENOSPC) std::fprintf(stderr, "amps::recc_fused: %s (bursts dropped, continuing)\n", amps_recc_strerror(rc));
            else if (rc != 0) { std::fprintf(stderr, "amps::recc_fused: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
            for (size_t i = 0; i < nrec; i++) {
                message_port_pub(pmt::mp("bursts"), pmt::mp(d_bursts.data() + i * AMPS_RECC_CAPTURE_SYMS, AMPS_RECC_CAPTURE_SYMS));
                message_port_pub(pmt::mp("records"), pmt::mp(&recs[i], sizeof(recs[i])));
            }
            done += n;
        }
        consume_each(noutput_items);
        return 0;
    }
};

recc_fused::sptr recc_fused::make(int samples_per_symbol, double xlate_rate_hz, double xlate_center_hz, int xlate_decim, double xlate_cutoff_hz, double xlate_width_hz)
{
    return gnuradio::get_initial_sptr(new recc_fused_impl(samples_per_symbol, xlate_rate_hz, xlate_center_hz, xlate_decim, xlate_cutoff_hz, xlate_width_hz));
}

} // namespace amps
} // namespace gr
