// gr::amps::recc -- same public interface as the reference block (include/amps/recc.h:15-29):
// a sync_block with one `unsigned char` input stream (one byte per Manchester symbol), no output
// stream, and a message output port "bursts" carrying 3374-byte blobs.  make() takes no parameters.
// The work is done by the MI355X path behind the C ABI (amps_recc_push_symbols).
#pragma once
#include <amps/api.h>

namespace gr {
namespace amps {

class AMPS_API recc : virtual public gr::sync_block {
public:
    typedef AMPS_SPTR<recc> sptr;
    static sptr make();
};

} // namespace amps
} // namespace gr
