// recc_wideband_impl.cc -- gr::amps::recc_wideband: one 30.72 Msps complex stream in, (channel, burst) and (channel, record) pairs out.
#include <amps/recc_wideband.h>
#include <cerrno>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "amps_recc.h"
#include <string>

namespace gr {
namespace amps {

class recc_wideband_impl : public recc_wideband {
    amps_recc_t *d_handle;
    std::vector<amps_recc_burst_t> d_recs;
    std::vector<unsigned char> d_bursts;
    static const int kMaxPush = 1 << 22;          // wideband samples per push (8192 frames)
    static const int kMaxRecs = 4096;
    bool d_bcast = false;                          // set_rccl: the stream comes from rank d_root by RCCL
    int d_root = 0, d_rank = 0;

public:
    recc_wideband_impl(int C, int first_bin, int slicer, int groups, int group)
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
        cfg.flags = AMPS_RECC_FLAG_KEEP_BURSTS | (slicer == 0 ? AMPS_RECC_FLAG_SLICER_ATAN : slicer == 1 ? AMPS_RECC_FLAG_SLICER_PRODUCT
                                                  : slicer == 2 ? AMPS_RECC_FLAG_SLICER_SINE : slicer == 3 ? AMPS_RECC_FLAG_SLICER_EXACT : 0u);
        cfg.wideband_groups = (uint32_t)groups;
        cfg.wideband_group = (uint32_t)group;
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

    void set_rccl(const std::string &id, int nranks, int rank, int root)
    {
        if (id.size() != AMPS_RECC_RCCL_ID_BYTES) throw std::runtime_error("amps::recc_wideband: the RCCL id is 128 bytes");
        int rc = amps_recc_rccl_init(d_handle, (const uint8_t *)id.data(), nranks, rank);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_wideband: rccl: ") + amps_recc_strerror(rc));
        d_bcast = true; d_root = root; d_rank = rank;
    }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        const float *in = (const float *)input_items[0];
        int done = 0;
        while (done < noutput_items) {
            int n = noutput_items - done;
            if (n > kMaxPush) n = kMaxPush;
            int rc;
            if (d_bcast)                            // every rank in step; only the root's items are read (staged to the device by the library)
                rc = amps_recc_push_wideband_bcast(d_handle, d_rank == d_root ? in + 2 * (size_t)done : nullptr, (size_t)n, AMPS_MEM_HOST, d_root);
            else rc = amps_recc_push_wideband(d_handle, in + 2 * (size_t)done, (size_t)n, AMPS_MEM_HOST);
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

recc_wideband::sptr recc_wideband::make(int n_channels, int first_bin, int slicer, int groups, int group)
{
    return gnuradio::get_initial_sptr(new recc_wideband_impl(n_channels, first_bin, slicer, groups, group));
}

std::string recc_wideband::rccl_unique_id()
{
    std::string id(AMPS_RECC_RCCL_ID_BYTES, '\0');
    int rc = amps_recc_rccl_unique_id((uint8_t *)&id[0]);
    if (rc != 0) throw std::runtime_error(std::string("amps::recc_wideband: rccl: ") + amps_recc_strerror(rc));
    return id;
}

} // namespace amps
} // namespace gr
