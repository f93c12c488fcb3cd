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
    int d_root = 0, d_rank = 0, d_mode = AMPS_RECC_DIST_BROADCAST;
    unsigned long long d_stream_samples = 0, d_paced_items = 0;   // non-root ranks: samples the root has distributed / items this rank's pacing input has offered
    bool d_ended = false;                          // the root's end of stream has been sent (root) / seen (the others)

public:
    recc_wideband_impl(int C, int first_bin, int slicer, int groups, int group, int decim)
        : gr::sync_block("recc_wideband", gr::io_signature::make(1, 1, 2 * sizeof(float)), gr::io_signature::make(0, 0, 0)), d_handle(nullptr),
          d_recs(kMaxRecs), d_bursts((size_t)kMaxRecs * AMPS_RECC_CAPTURE_SYMS)
    {
        amps_recc_cfg_t cfg = {};
        cfg.struct_size = sizeof(cfg);
        cfg.n_channels = (uint32_t)C;
        cfg.samples_per_symbol = 0;                // what goes with the decimation: 3 (60 ksps per channel, D = 512) or 2 (40 ksps, D = 768)
        cfg.max_samples_per_push = kMaxPush / 512 + 72;
        cfg.max_bursts = kMaxRecs;
        cfg.device = -1;
        cfg.flags = AMPS_RECC_FLAG_KEEP_BURSTS | (slicer == 0 ? AMPS_RECC_FLAG_SLICER_ATAN : slicer == 1 ? AMPS_RECC_FLAG_SLICER_PRODUCT
                                                  : slicer == 2 ? AMPS_RECC_FLAG_SLICER_SINE : slicer == 3 ? AMPS_RECC_FLAG_SLICER_EXACT : 0u);
        cfg.wideband_groups = (uint32_t)groups;
        cfg.wideband_group = (uint32_t)group;
        cfg.wideband_channels = 1024;
        cfg.wideband_decim = (uint32_t)decim;     // 0 = amps_recc_default_wideband_decim()
        cfg.wideband_taps_per_branch = 8;
        cfg.wideband_first_channel = (uint32_t)first_bin;
        int rc = amps_recc_create(&d_handle, &cfg);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_wideband: ") + amps_recc_strerror(rc));
        message_port_register_out(pmt::mp("bursts"));
        message_port_register_out(pmt::mp("records"));
    }
    ~recc_wideband_impl() { amps_recc_destroy(d_handle); }

    void set_rccl(const std::string &id, int nranks, int rank, int root, int mode)
    {
        if (id.size() != AMPS_RECC_RCCL_ID_BYTES) throw std::runtime_error("amps::recc_wideband: the RCCL id is 128 bytes");
        if (mode != AMPS_RECC_DIST_BROADCAST && mode != AMPS_RECC_DIST_SCATTER_ALLGATHER) throw std::runtime_error("amps::recc_wideband: unknown distribution mode");
        int rc = amps_recc_rccl_init(d_handle, (const uint8_t *)id.data(), nranks, rank);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_wideband: rccl: ") + amps_recc_strerror(rc));
        d_bcast = true; d_root = root; d_rank = rank; d_mode = mode;
    }

    // one block through the seam (alone, or as this rank's part of one collective) + its records out on the ports; false = stop the flow graph
    bool push_and_publish(const float *in, size_t n)
    {
        int rc;
        size_t pushed = n;
        if (d_bcast) {
            // every rank in step, ONE collective per call; only the root's items are read (staged to the device by the library) and only
            // the root's n counts -- it travels in the header of the collective (the schedulers of the ranks' flow graphs do not agree on
            // noutput_items)
            const bool root = d_rank == d_root;
            rc = amps_recc_push_wideband_dist(d_handle, root ? in : nullptr, root ? n : 0, AMPS_MEM_HOST, d_root, d_mode, &pushed);
            d_stream_samples += pushed;
        } else rc = amps_recc_push_wideband(d_handle, in, n, AMPS_MEM_HOST);
        if (rc == -ENODATA) { d_ended = true; return false; }   // the root's stream has ended: a clean WORK_DONE, the communicator is left alone
        if (rc != 0) { std::fprintf(stderr, "amps::recc_wideband: %s\n", amps_recc_strerror(rc)); return leave(); }
        size_t nrec = 0;
        rc = amps_recc_drain_bursts(d_handle, d_recs.data(), d_bursts.data(), kMaxRecs, &nrec);
        // -ENOSPC: more bursts than the list holds were found; the ones that fit are returned and the list recovers on the
        // next push -- a recoverable condition must not end the flow graph
        if (rc == -ENOSPC) std::fprintf(stderr, "amps::recc_wideband: %s (bursts dropped, continuing)\n", amps_recc_strerror(rc));
        else if (rc != 0) { std::fprintf(stderr, "amps::recc_wideband: %s\n", amps_recc_strerror(rc)); return leave(); }
        for (size_t i = 0; i < nrec; i++) {
            const pmt::pmt_t ch = pmt::from_long((long)d_recs[i].channel);
            message_port_pub(pmt::mp("bursts"), pmt::cons(ch, pmt::mp(d_bursts.data() + i * AMPS_RECC_CAPTURE_SYMS, AMPS_RECC_CAPTURE_SYMS)));
            message_port_pub(pmt::mp("records"), pmt::cons(ch, pmt::mp(&d_recs[i], sizeof(d_recs[i]))));
        }
        return true;
    }
    // The flow graph has stopped (the ranks' sources end at different times).  The root tells the others -- one header
    // exchange, -ENODATA on every rank, no data collective; the others keep joining the root's collectives (and publishing their records)
    // until they see it, so the root's last blocks are not lost on ranks whose pacing source ended first.  A root that died instead
    // is what the bounded waits are for.
    bool stop() override
    {
        if (d_bcast && !d_ended) {
            if (d_rank == d_root) { (void)amps_recc_push_wideband_dist(d_handle, nullptr, 0, AMPS_MEM_HOST, d_root, d_mode, nullptr); d_ended = true; }
            else while (!d_ended && push_and_publish(nullptr, 0)) { }
        }
        return true;
    }
    // this rank stops: the others must not wait for it beyond their bound
    bool leave()
    {
        if (d_bcast) (void)amps_recc_rccl_abort(d_handle);
        return false;
    }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        const float *in = (const float *)input_items[0];
        if (d_bcast && d_rank != d_root) {
            // A non-root rank: its items are pacing only.  The ranks stay in step by STREAM POSITION, not by call or item counts (which
            // the schedulers of different processes do not share): this rank joins collectives -- each of whatever size the root
            // announces in its header -- until the root's stream has covered the items its own pacing input has offered so far.
            d_paced_items += (unsigned long long)noutput_items;
            while (d_stream_samples < d_paced_items)
                if (!push_and_publish(nullptr, 0)) return WORK_DONE;
            consume_each(noutput_items);
            return 0;
        }
        int done = 0;
        while (done < noutput_items) {
            int n = noutput_items - done;
            if (n > kMaxPush) n = kMaxPush;
            if (!push_and_publish(in + 2 * (size_t)done, (size_t)n)) return WORK_DONE;
            done += n;
        }
        consume_each(noutput_items);
        return 0;
    }
};

recc_wideband::sptr recc_wideband::make(int n_channels, int first_bin, int slicer, int groups, int group, int decim)
{
    return gnuradio::get_initial_sptr(new recc_wideband_impl(n_channels, first_bin, slicer, groups, group, decim));
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
