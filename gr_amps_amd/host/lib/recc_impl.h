// recc_impl.h -- private implementation of gr::amps::recc over the MI355X C ABI.
// Mirrors the shape of the reference's lib/recc_impl.h:28-52 (private impl class, public make()).
#pragma once
#include <amps/recc.h>
#include "amps_recc.h"

namespace gr {
namespace amps {

class recc_impl : public recc {
private:
    amps_recc_t *d_handle;          // owns the per-channel symbol buffer + trigger state on the device
    std::vector<unsigned char> d_burst;

public:
    recc_impl();
    ~recc_impl();
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items);
};

} // namespace amps
} // namespace gr
