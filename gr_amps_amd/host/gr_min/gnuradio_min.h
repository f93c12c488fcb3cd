// gnuradio_min.h -- the handful of GNU Radio 3.7 runtime types the RECC blocks touch, so that the
// host-side mirror of gr::amps::recc / gr::amps::recc_decode builds and runs on a box without GNU
// Radio (neither this image nor the GPU box has it).  It is OUR host layer, not a stand-in used to
// build the reference: nothing from /root/reference is compiled against it.
//
// Names and call shapes follow GNU Radio so that the block sources (host/lib/*.cc) also compile
// against the real headers when AMPS_WITH_GNURADIO is defined:
//   gr::io_signature::make, gr::sync_block::work, gr::block::general_work, consume_each,
//   message_port_register_in/out, message_port_pub, set_msg_handler, gnuradio::get_initial_sptr,
//   pmt::mp / blob / tuple / from_long / from_bool / from_uint64 / cons / make_dict / init_u8vector.
// Message delivery here is synchronous on the publisher's thread (GNU Radio queues to the
// receiving block's thread; ordering per port is the same).
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
typedef std::vector<int> gr_vector_int;

namespace pmt {
struct pmt_base;
typedef std::shared_ptr<pmt_base> pmt_t;
struct pmt_base {
    enum kind_t { NIL, SYMBOL, BLOB, LONG, UINT64, BOOL, TUPLE, PAIR, DICT, U8VECTOR } kind = NIL;
    long l = 0;
    uint64_t u = 0;
    bool b = false;
    std::string s;
    std::vector<uint8_t> bytes;
    std::vector<pmt_t> items;
};
inline pmt_t make(pmt_base::kind_t k) { auto p = std::make_shared<pmt_base>(); p->kind = k; return p; }
inline pmt_t mp(const char *s) { auto p = make(pmt_base::SYMBOL); p->s = s; return p; }
inline pmt_t mp(const std::string &s) { return mp(s.c_str()); }
inline pmt_t mp(const void *data, size_t len) // blob copy, like pmt::mp(ptr, len) -> make_blob
{
    auto p = make(pmt_base::BLOB);
    p->bytes.assign((const uint8_t *)data, (const uint8_t *)data + len);
    return p;
}
inline pmt_t intern(const std::string &s) { return mp(s); }
inline std::string symbol_to_string(const pmt_t &p) { return p->s; }
inline bool is_blob(const pmt_t &p) { return p && p->kind == pmt_base::BLOB; }
inline size_t blob_length(const pmt_t &p) { return p->bytes.size(); }
inline const void *blob_data(const pmt_t &p) { return p->bytes.data(); }
inline pmt_t from_long(long v) { auto p = make(pmt_base::LONG); p->l = v; return p; }
inline long to_long(const pmt_t &p) { return p->l; }
inline pmt_t from_uint64(uint64_t v) { auto p = make(pmt_base::UINT64); p->u = v; return p; }
inline uint64_t to_uint64(const pmt_t &p) { return p->u; }
inline pmt_t from_bool(bool v) { auto p = make(pmt_base::BOOL); p->b = v; return p; }
inline bool to_bool(const pmt_t &p) { return p->b; }
inline bool is_tuple(const pmt_t &p) { return p && p->kind == pmt_base::TUPLE; }
inline pmt_t make_tuple(const pmt_t &a, const pmt_t &b, const pmt_t &c)
{
    auto p = make(pmt_base::TUPLE); p->items = { a, b, c }; return p;
}
inline pmt_t make_tuple(const pmt_t &a, const pmt_t &b, const pmt_t &c, const pmt_t &d)
{
    auto p = make(pmt_base::TUPLE); p->items = { a, b, c, d }; return p;
}
inline pmt_t tuple_ref(const pmt_t &t, size_t k) { return t->items.at(k); }
inline size_t length(const pmt_t &t) { return t->items.size(); }
inline pmt_t make_dict() { return make(pmt_base::DICT); }
inline pmt_t cons(const pmt_t &a, const pmt_t &b) { auto p = make(pmt_base::PAIR); p->items = { a, b }; return p; }
inline pmt_t car(const pmt_t &p) { return p->items.at(0); }
inline pmt_t cdr(const pmt_t &p) { return p->items.at(1); }
inline pmt_t init_u8vector(size_t n, const uint8_t *data)
{
    auto p = make(pmt_base::U8VECTOR); p->bytes.assign(data, data + n); return p;
}
inline const uint8_t *u8vector_elements(const pmt_t &p, size_t &len) { len = p->bytes.size(); return p->bytes.data(); }
} // namespace pmt

namespace gr {

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_item)
    {
        auto s = std::make_shared<io_signature>();
        s->min_streams = min_streams; s->max_streams = max_streams; s->item_size = sizeof_item;
        return s;
    }
    int min_streams = 0, max_streams = 0, item_size = 0;
};

class basic_block : public std::enable_shared_from_this<basic_block> {
public:
    typedef std::function<void(pmt::pmt_t)> msg_handler_t;
    basic_block(const std::string &name, io_signature::sptr in, io_signature::sptr out)
        : d_name(name), d_in(in), d_out(out) {}
    basic_block() {} // for pure-interface subclasses that inherit virtually (as in GNU Radio)
    virtual ~basic_block() {}
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    void message_port_register_in(const pmt::pmt_t &port) { d_in_ports[pmt::symbol_to_string(port)]; }
    void message_port_register_out(const pmt::pmt_t &port) { d_subs[pmt::symbol_to_string(port)]; }
    bool has_msg_port_out(const std::string &p) const { return d_subs.count(p) != 0; }
    bool has_msg_port_in(const std::string &p) const { return d_in_ports.count(p) != 0; }
    template <typename F> void set_msg_handler(const pmt::pmt_t &port, F f) { d_in_ports[pmt::symbol_to_string(port)] = f; }
    void message_port_pub(const pmt::pmt_t &port, const pmt::pmt_t &msg)
    {
        auto it = d_subs.find(pmt::symbol_to_string(port));
        if (it == d_subs.end()) return;
        for (auto &sub : it->second) sub(msg);
    }
    // flow-graph edge: src.port -> handler of dst.port (msg_connect in a GR top_block)
    void subscribe(const std::string &port, msg_handler_t h) { d_subs[port].push_back(h); }
    void dispatch(const std::string &port, const pmt::pmt_t &msg)
    {
        auto it = d_in_ports.find(port);
        if (it != d_in_ports.end() && it->second) it->second(msg);
    }
protected:
    std::string d_name;
    io_signature::sptr d_in, d_out;
    std::map<std::string, msg_handler_t> d_in_ports;
    std::map<std::string, std::vector<msg_handler_t>> d_subs;
};

class block : public basic_block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : basic_block(name, in, out) {}
    block() {}
    virtual void forecast(int, gr_vector_int &) {}
    virtual bool start() { return true; }      // gr::block::start / stop: called by the scheduler when the flow graph starts / has stopped
    virtual bool stop() { return true; }
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
    void consume_each(int n) { d_consumed += n; }
    long consumed() const { return d_consumed; }
private:
    long d_consumed = 0;
};

class sync_block : public block {
public:
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : block(name, in, out) {}
    sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override
    {
        return work(noutput_items, input_items, output_items);
    }
};

inline void msg_connect(const std::shared_ptr<basic_block> &src, const std::string &sport,
                        const std::shared_ptr<basic_block> &dst, const std::string &dport)
{
    std::weak_ptr<basic_block> w = dst;
    src->subscribe(sport, [w, dport](pmt::pmt_t m) { if (auto d = w.lock()) d->dispatch(dport, m); });
}

} // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
} // namespace gnuradio
