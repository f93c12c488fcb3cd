// gr::amps::recc_decode -- same public interface as the reference block
// (include/amps/recc_decode.h:15-29): a gr::block without streams; message input "bursts";
// message outputs focc_words, fvc_words, audio_mute, fvc_mute, command_out
// (lib/recc_decode_impl.cc:38-46).  Decoding runs on the MI355X (amps_recc_decode_bursts).
#pragma once
#include <amps/api.h>

namespace gr {
namespace amps {

class AMPS_API recc_decode : virtual public gr::block {
public:
    typedef AMPS_SPTR<recc_decode> sptr;
    static sptr make();
};

} // namespace amps
} // namespace gr
