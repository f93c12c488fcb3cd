// recc_impl.cc -- gr::amps::recc: byte symbols in, "bursts" blobs out; same contract as the reference
// block (lib/recc_impl.cc:67-145) with the buffer state machine and trigger search running on the
// MI355X (amps_recc_push_symbols).  One block instance = one RECC channel, like the reference.
#include "recc_impl.h"
#include <cstdio>
#include <stdexcept>

namespace gr {
namespace amps {

recc::sptr recc::make() { return gnuradio::get_initial_sptr(new recc_impl()); }   // lib/recc_impl.cc:30-33

recc_impl::recc_impl()
    : gr::sync_block("recc", gr::io_signature::make(1, 1, sizeof(unsigned char)),   // lib/recc_impl.cc:71-73
                     gr::io_signature::make(0, 0, 0)),
      d_handle(nullptr), d_burst(AMPS_RECC_CAPTURE_SYMS)
{
    amps_recc_cfg_t cfg = {};
    cfg.struct_size = sizeof(cfg);
    cfg.n_channels = 1;
    cfg.max_bursts = 4;
    cfg.device = -1;
    int rc = amps_recc_create(&d_handle, &cfg);
    if (rc != 0)   // the reference cannot fail here; without a GPU this block has nothing to run on
        throw std::runtime_error(std::string("amps::recc: ") + amps_recc_strerror(rc));
    message_port_register_out(pmt::mp("bursts"));                                    // lib/recc_impl.cc:82
}

recc_impl::~recc_impl() { amps_recc_destroy(d_handle); }

int recc_impl::work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    const unsigned char *in = (const unsigned char *)input_items[0];
    if (noutput_items < 1) {                                                         // lib/recc_impl.cc:99-102
        std::printf("XXX noutput_items is %d\n", noutput_items);
        return 0;
    }
    // GNU Radio never hands more than the buffer allows; chunk defensively at the reference's own limit
    int done = 0;
    while (done < noutput_items) {
        int n = noutput_items - done;
        if (n > AMPS_RECC_MAX_WORK_ITEMS) n = AMPS_RECC_MAX_WORK_ITEMS;
        uint32_t chan = 0;
        size_t nout = 0;
        int rc = amps_recc_push_symbols(d_handle, in + done, (size_t)n, n, AMPS_MEM_HOST, d_burst.data(), &chan, 1, &nout);
        if (rc != 0) { std::fprintf(stderr, "amps::recc: %s\n", amps_recc_strerror(rc)); return WORK_DONE; }
        if (nout == 1) message_port_pub(pmt::mp("bursts"), pmt::mp(d_burst.data(), AMPS_RECC_CAPTURE_SYMS));  // :126
        done += n;
    }
    consume_each(noutput_items);                                                     // lib/recc_impl.cc:113
    return 0;                                                                        // lib/recc_impl.cc:144
}

} // namespace amps
} // namespace gr
