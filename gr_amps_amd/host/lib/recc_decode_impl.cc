// recc_decode_impl.cc -- gr::amps::recc_decode: "bursts" blobs in, control messages out; same ports
// and payload shapes as the reference (lib/recc_decode_impl.cc:32-47, 181-272).  The Manchester
// decode, the 35 BCH decodes, the field parse and the dispatch run on the MI355X
// (amps_recc_decode_bursts); the reply words come from amps_recc_reply_words.
#include "recc_decode_impl.h"
#ifdef AMPS_WITH_GNURADIO
#include <boost/bind.hpp>
#endif
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace gr {
namespace amps {

recc_decode::sptr recc_decode::make() { return gnuradio::get_initial_sptr(new recc_decode_impl()); }  // :22-27

recc_decode_impl::recc_decode_impl()
    : gr::block("recc_decode", gr::io_signature::make(0, 0, 0), gr::io_signature::make(0, 0, 0)), d_handle(nullptr)  // :34-36
{
    amps_recc_cfg_t cfg = {};
    cfg.struct_size = sizeof(cfg);
    cfg.n_channels = 1;
    cfg.max_bursts = 4;
    cfg.device = -1;
    int rc = amps_recc_create(&d_handle, &cfg);
    if (rc != 0) throw std::runtime_error(std::string("amps::recc_decode: ") + amps_recc_strerror(rc));
    message_port_register_in(pmt::mp("bursts"));                                        // :38
    message_port_register_in(pmt::mp("records"));                                       // extra: from recc_fused
#ifdef AMPS_WITH_GNURADIO
    // GNU Radio 3.7's msg_handler_t is a boost::function: the reference's own registration form (:39-41)
    set_msg_handler(pmt::mp("bursts"), boost::bind(&recc_decode_impl::bursts_message, this, _1));
    set_msg_handler(pmt::mp("records"), boost::bind(&recc_decode_impl::records_message, this, _1));
#else
    set_msg_handler(pmt::mp("bursts"), [this](pmt::pmt_t m) { bursts_message(m); });    // :39-41
    set_msg_handler(pmt::mp("records"), [this](pmt::pmt_t m) { records_message(m); });
#endif
    message_port_register_out(pmt::mp("focc_words"));                                   // :42-46
    message_port_register_out(pmt::mp("fvc_words"));
    message_port_register_out(pmt::mp("audio_mute"));
    message_port_register_out(pmt::mp("fvc_mute"));
    message_port_register_out(pmt::mp("command_out"));
}

recc_decode_impl::~recc_decode_impl() { amps_recc_destroy(d_handle); }

void recc_decode_impl::bursts_message(pmt::pmt_t msg)                                   // :81
{
    if (!pmt::is_blob(msg) || pmt::blob_length(msg) != AMPS_RECC_CAPTURE_SYMS) {
        std::printf("recc_decode: ignoring a message that is not a %d-byte burst blob\n", AMPS_RECC_CAPTURE_SYMS);
        return;
    }
    amps_recc_burst_t rec;
    int rc = amps_recc_decode_bursts(d_handle, (const uint8_t *)pmt::blob_data(msg), 1, AMPS_MEM_HOST, nullptr, &rec);
    if (rc != 0) { std::fprintf(stderr, "amps::recc_decode: %s\n", amps_recc_strerror(rc)); return; }
    publish_reply(rec);
}

void recc_decode_impl::records_message(pmt::pmt_t msg)
{
    if (!pmt::is_blob(msg) || pmt::blob_length(msg) != sizeof(amps_recc_burst_t)) return;
    amps_recc_burst_t rec;
    std::memcpy(&rec, pmt::blob_data(msg), sizeof(rec));
    publish_reply(rec);
}

void recc_decode_impl::publish_reply(const amps_recc_burst_t &rec)
{
    switch (rec.msg_class) {   // the log lines of the reference, :108-168
    case AMPS_MSG_INVALID_WORD_A: std::printf("DEBUG: got a burst with an invalid Word A\n"); return;
    case AMPS_MSG_E_ZERO: std::printf("WARNING: got a RECC message with E=0; not sure what this is\n"); return;
    case AMPS_MSG_BAD_NAWC: std::printf("WARNING: invalid NAWC value in RECC origination\n"); return;
    case AMPS_MSG_UNKNOWN:
        std::printf("WARNING: got unknown RECC message: ORDER 0x%x  ORDQ 0x%x  MSG_TYPE 0x%x\n", rec.b_ORDER, rec.b_ORDQ, rec.b_MSG_TYPE);
        return;
    case AMPS_MSG_REGISTRATION: std::printf("DEBUG: got registration from MIN=%s\n", rec.min); break;
    case AMPS_MSG_PAGE_RESPONSE: std::printf("DEBUG: got a response from MIN=%s\n", rec.min); break;
    case AMPS_MSG_ORIGINATION: std::printf("DEBUG: origination: MIN=%s ESN=%x dialed %s\n", rec.min, rec.esn, rec.dialed); break;
    default: return;
    }
    amps_recc_reply_t r;
    if (amps_recc_reply_words(&rec, &r) != 0) return;
    if (r.has_focc)    // tuple(long stream, long nwords, blob28, blob28): :188, :209, :261
        message_port_pub(pmt::mp("focc_words"),
                         pmt::make_tuple(pmt::from_long(r.focc_stream), pmt::from_long(r.focc_nwords), pmt::mp(r.focc_word1, 28),
                                         pmt::mp(r.focc_word2, 28)));
    if (r.has_fvc)     // tuple(long 1, blob28, uint64 35): :215
        message_port_pub(pmt::mp("fvc_words"), pmt::make_tuple(pmt::from_long(r.fvc_count), pmt::mp(r.fvc_word1, 28), pmt::from_uint64(r.fvc_repeat)));
    if (r.has_mutes) { // :219-220, :265-266
        message_port_pub(pmt::mp("fvc_mute"), pmt::from_bool(r.fvc_mute != 0));
        message_port_pub(pmt::mp("audio_mute"), pmt::from_bool(r.audio_mute != 0));
    }
    if (r.has_command) // PDU cons(dict, u8vector("page <digits>")): :268-271
        message_port_pub(pmt::mp("command_out"), pmt::cons(pmt::make_dict(), pmt::init_u8vector(std::strlen(r.command), (const uint8_t *)r.command)));
}

void recc_decode_impl::forecast(int, gr_vector_int &) {}                               // :281-285

int recc_decode_impl::general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &)
{
    consume_each(noutput_items);                                                        // :287-296 (no-op block)
    return noutput_items;
}

} // namespace amps
} // namespace gr
