// recc_bank_impl.cc -- gr::amps::recc_bank: C byte-symbol streams in, (channel, burst blob) pairs out; one device launch per work().
#include <amps/recc_bank.h>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "amps_recc.h"

namespace gr {
namespace amps {

class recc_bank_impl : public recc_bank {
    amps_recc_t *d_handle;
    int d_C;
    std::vector<unsigned char> d_stage, d_bursts;
    std::vector<uint32_t> d_chan;

public:
    explicit recc_bank_impl(int C)
        : gr::sync_block("recc_bank", gr::io_signature::make(C, C, sizeof(unsigned char)), gr::io_signature::make(0, 0, 0)),
          d_handle(nullptr), d_C(C), d_bursts((size_t)C * AMPS_RECC_CAPTURE_SYMS), d_chan((size_t)C)
    {
        if (C < 1) throw std::invalid_argument("amps::recc_bank: n_channels < 1");
        amps_recc_cfg_t cfg = {};
        cfg.struct_size = sizeof(cfg);
        cfg.n_channels = (uint32_t)C;
        cfg.max_bursts = (uint32_t)C;                          // a channel publishes at most one burst per work() call
        cfg.device = -1;
        int rc = amps_recc_create(&d_handle, &cfg);
        if (rc != 0) throw std::runtime_error(std::string("amps::recc_bank: ") + amps_recc_strerror(rc));
        message_port_register_out(pmt::mp("bursts"));
    }
    ~recc_bank_impl() { amps_recc_destroy(d_handle); }

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        if (noutput_items < 1) return 0;                                               // lib/recc_impl.cc:99-102
        int done = 0;
        while (done < noutput_items) {
            int n = noutput_items - done;
            if (n > AMPS_RECC_MAX_WORK_ITEMS) n = AMPS_RECC_MAX_WORK_ITEMS;            // lib/recc_impl.cc:103
            // GNU Radio hands one pointer per stream: gather the C chunks into one [C][n] block (the C ABI's layout)
            d_stage.resize((size_t)d_C * n);
            for (int c = 0; c < d_C; c++) std::memcpy(&d_stage[(size_t)c * n], (const unsigned char *)input_items[c] + done, (size_t)n);
            size_t nout = 0;
            int rc = amps_recc_push_symbols(d_handle, d_stage.data(), (size_t)n, n, AMPS_MEM_HOST, d_bursts.data(), d_chan.data(), (size_t)d_C, &nout);
            if (rc == -ENOSPC) std::fprintf(stderr, "amps::recc_bank: %s (bursts dropped, continuing)\n", amps_recc_strerror(rc));   // recoverable
            else if (rc != 0) { std::fprintf(stderr, "amps::recc_bank: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
            for (size_t i = 0; i < nout; i++)
                message_port_pub(pmt::mp("bursts"), pmt::cons(pmt::from_long((long)d_chan[i]),
                                                              pmt::mp(d_bursts.data() + i * AMPS_RECC_CAPTURE_SYMS, AMPS_RECC_CAPTURE_SYMS)));
            done += n;
        }
        consume_each(noutput_items);
        return 0;
    }
};

recc_bank::sptr recc_bank::make(int n_channels) { return gnuradio::get_initial_sptr(new recc_bank_impl(n_channels)); }

} // namespace amps
} // namespace gr
