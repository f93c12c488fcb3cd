// recc_decode_impl.h -- private implementation of gr::amps::recc_decode over the MI355X C ABI
// (shape of the reference's lib/recc_decode_impl.h:20-43).
#pragma once
#include <amps/recc_decode.h>
#include "amps_recc.h"

namespace gr {
namespace amps {

class recc_decode_impl : public recc_decode {
private:
    amps_recc_t *d_handle;
    void publish_reply(const amps_recc_burst_t &rec);

public:
    recc_decode_impl();
    ~recc_decode_impl();
    void forecast(int noutput_items, gr_vector_int &ninput_items_required);
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items);
    void bursts_message(pmt::pmt_t msg);
    void records_message(pmt::pmt_t msg);   // already decoded records from gr::amps::recc_fused
};

} // namespace amps
} // namespace gr
