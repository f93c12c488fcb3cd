// recctest.cc -- config 0 plumbing: the message part of grc/recctest.grc (connections :3238-3274)
// as a stand-alone program.  Modes:
//   recctest syms <file.u8>  [chunk]   u8 0/1 symbol file -> amps.recc -> amps.recc_decode
//   recctest iq   <file.fc32> [chunk]  200 ksps interleaved fc32 -> amps.recc_fused -> amps.recc_decode
//   recctest iqb  <file.fc32> [chunk]  the same through recc_fused's "bursts" port (the 3374-byte blob amps_recc publishes) instead of "records"
//   recctest bank <file.u8>  [chunk] [C]  C channels in ONE gr::amps::recc_bank block: channel c = the symbol file delayed by 37 c symbols;
//                                      every line is prefixed with the channel the burst came from
//   recctest wide <file.fc32> [chunk] [slicer] [decim]  one 30.72 Msps wideband capture -> gr::amps::recc_wideband (832 channels from bin 96); every
//                                      burst's lines are prefixed with its channel; decoded through the "bursts" port
//   recctest widerank <file.fc32> <chunk> <idfile> <nranks> <rank> [mode]   ONE rank of the same band over `nranks` processes (2, 4 or 8; one per GPU of a
//                                      node): gr::amps::recc_wideband::make(832, 96, -1, nranks, rank) + set_rccl -- rank 0 owns the capture and the
//                                      library distributes it (mode 0 = ncclBroadcast, 1 = scatter + all-gather); the other ranks' items only pace
//                                      them (they may use any chunk).  The 128-byte communicator id travels through <idfile> (rank 0 writes it).
//                                      Each rank prints the bursts of ITS channel group; the union is what `recctest wide` prints
//   recctest raw  <file.fc32> [chunk] [center_hz] [cutoff_hz]   the flow graph's own capture format (grc/recctest.grc:591): 400 ksps fc32,
//                                      channel at center_hz (default +160 kHz, :889-937) -> channel filter + fused chain on the GPU;
//                                      cutoff_hz (default 0 = the flow graph's 10 kHz) widens the channel filter for mobiles off their carrier
// Every message published on recc_decode's output ports is printed as one text line, which is what
// tests/test_gpu_host_blocks.py compares with the oracle.
#include <amps/recc.h>
#include <amps/recc_decode.h>
#include <amps/recc_bank.h>
#include <amps/recc_fused.h>
#include <amps/recc_wideband.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

static std::string bits(const pmt::pmt_t &blob)
{
    std::string s;
    const uint8_t *p = (const uint8_t *)pmt::blob_data(blob);
    for (size_t i = 0; i < pmt::blob_length(blob); i++) s += p[i] ? '1' : '0';
    return s;
}

namespace {
struct sink : gr::block {   // prints what ampsbs.grc would route to focc / fvc / mutes / command_processor
    sink() : gr::block("sink", gr::io_signature::make(0, 0, 0), gr::io_signature::make(0, 0, 0))
    {
        const char *ports[] = { "focc_words", "fvc_words", "audio_mute", "fvc_mute", "command_out" };
        for (const char *p : ports) {
            message_port_register_in(pmt::mp(p));
            std::string name = p;
            set_msg_handler(pmt::mp(p), [name](pmt::pmt_t m) {
                if (name == "focc_words")
                    std::printf("MSG focc_words stream=%ld n=%ld w1=%s w2=%s\n", pmt::to_long(pmt::tuple_ref(m, 0)), pmt::to_long(pmt::tuple_ref(m, 1)),
                                bits(pmt::tuple_ref(m, 2)).c_str(), bits(pmt::tuple_ref(m, 3)).c_str());
                else if (name == "fvc_words")
                    std::printf("MSG fvc_words n=%ld w1=%s repeat=%llu\n", pmt::to_long(pmt::tuple_ref(m, 0)), bits(pmt::tuple_ref(m, 1)).c_str(),
                                (unsigned long long)pmt::to_uint64(pmt::tuple_ref(m, 2)));
                else if (name == "command_out") {
                    size_t n = 0;
                    const uint8_t *p = pmt::u8vector_elements(pmt::cdr(m), n);
                    std::printf("MSG command_out %.*s\n", (int)n, (const char *)p);
                } else
                    std::printf("MSG %s %d\n", name.c_str(), pmt::to_bool(m) ? 1 : 0);
            });
        }
    }
    int general_work(int n, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) override { return n; }
};
} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s syms|iq|iqb|raw|bank|wide <file> [chunk] [center_hz | C | slicer]\n", argv[0]); return 2; }
    const std::string mode = argv[1];
    const int chunk = argc > 3 ? std::atoi(argv[3]) : 4096;
    std::ifstream f(argv[2], std::ios::binary);
    if (!f) { std::perror(argv[2]); return 2; }
    std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        auto dec = gr::amps::recc_decode::make();
        auto snk = std::make_shared<sink>();
        for (const char *p : { "focc_words", "fvc_words", "audio_mute", "fvc_mute", "command_out" }) gr::msg_connect(dec, p, snk, p);
        gr_vector_void_star outs;
        if (mode == "bank") {
            const int C = argc > 4 ? std::atoi(argv[4]) : 8;
            auto src = gr::amps::recc_bank::make(C);
            // demultiplex: (channel, blob) -> print the channel, hand the blob to the one recc_decode
            struct demux : gr::block {
                std::shared_ptr<gr::basic_block> dec;
                demux() : gr::block("demux", gr::io_signature::make(0, 0, 0), gr::io_signature::make(0, 0, 0))
                {
                    message_port_register_in(pmt::mp("bursts"));
                    set_msg_handler(pmt::mp("bursts"), [this](pmt::pmt_t m) {
                        std::printf("MSG channel %ld\n", pmt::to_long(pmt::car(m)));
                        dec->dispatch("bursts", pmt::cdr(m));
                    });
                }
                int general_work(int n, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) override { return n; }
            };
            auto dm = std::make_shared<demux>();
            dm->dec = dec;
            gr::msg_connect(src, "bursts", dm, "bursts");
            std::vector<std::vector<char>> chans((size_t)C);
            for (int c = 0; c < C; c++) {                       // channel c: 37 c idle symbols in front of the file
                chans[c].assign((size_t)37 * c, 0);
                chans[c].insert(chans[c].end(), data.begin(), data.end());
                chans[c].resize(data.size() + (size_t)37 * C, 0);
            }
            const size_t total = data.size() + (size_t)37 * C;
            for (size_t off = 0; off < total; off += (size_t)chunk) {
                int n = (int)std::min<size_t>((size_t)chunk, total - off);
                gr_vector_const_void_star ins;
                for (int c = 0; c < C; c++) ins.push_back(chans[c].data() + off);
                if (src->work(n, ins, outs) != 0) return 1;
            }
        } else if (mode == "wide" || mode == "widerank") {
            const bool ranks = mode == "widerank";
            if (ranks && argc < 7) { std::fprintf(stderr, "usage: %s widerank <file> <chunk> <idfile> <nranks> <rank> [mode]\n", argv[0]); return 2; }
            const int nranks = ranks ? std::atoi(argv[5]) : 0, rank = ranks ? std::atoi(argv[6]) : 0;
            auto src = ranks ? gr::amps::recc_wideband::make(832, 96, -1, nranks, rank) : gr::amps::recc_wideband::make(832, 96, argc > 4 ? std::atoi(argv[4]) : -1, 0, 0, argc > 5 ? std::atoi(argv[5]) : 0);
            if (ranks) {
                // the control plane is the application's: here, a file
                std::string id;
                const std::string idfile = argv[4];
                if (rank == 0) {
                    id = gr::amps::recc_wideband::rccl_unique_id();
                    std::ofstream o(idfile + ".tmp", std::ios::binary);
                    o.write(id.data(), (std::streamsize)id.size());
                    o.close();
                    std::rename((idfile + ".tmp").c_str(), idfile.c_str());
                } else {
                    for (int tries = 0; tries < 12000 && id.size() != 128; tries++) {
                        std::ifstream i(idfile, std::ios::binary);
                        if (i) id.assign((std::istreambuf_iterator<char>(i)), std::istreambuf_iterator<char>());
                        if (id.size() != 128) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                    }
                }
                src->set_rccl(id, nranks, rank, 0, argc > 7 ? std::atoi(argv[7]) : 0);
            }
            struct demux : gr::block {
                std::shared_ptr<gr::basic_block> dec;
                demux() : gr::block("demux", gr::io_signature::make(0, 0, 0), gr::io_signature::make(0, 0, 0))
                {
                    message_port_register_in(pmt::mp("bursts"));
                    set_msg_handler(pmt::mp("bursts"), [this](pmt::pmt_t m) {
                        std::printf("MSG channel %ld\n", pmt::to_long(pmt::car(m)));
                        dec->dispatch("bursts", pmt::cdr(m));
                    });
                }
                int general_work(int n, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) override { return n; }
            };
            auto dm = std::make_shared<demux>();
            dm->dec = dec;
            gr::msg_connect(src, "bursts", dm, "bursts");
            std::vector<char> tail((size_t)64 * 768 * 8, 0);          // silence: flushes the frames the fused form holds back (64 frames of at most 768 samples)
            data.insert(data.end(), tail.begin(), tail.end());
            const size_t ns = data.size() / 8;
            for (size_t off = 0; off < ns; off += (size_t)chunk) {
                int n = (int)std::min<size_t>((size_t)chunk, ns - off);
                gr_vector_const_void_star ins = { data.data() + 8 * off };
                if (src->work(n, ins, outs) != 0) return 1;
            }
            src->stop();                                              // as the scheduler does: the root announces its end of stream, the others join until they see it
        } else if (mode == "syms") {
            auto src = gr::amps::recc::make();
            gr::msg_connect(src, "bursts", dec, "bursts");
            for (size_t off = 0; off < data.size(); off += (size_t)chunk) {
                int n = (int)std::min<size_t>((size_t)chunk, data.size() - off);
                gr_vector_const_void_star ins = { data.data() + off };
                if (src->work(n, ins, outs) != 0) return 1;
            }
        } else {
            const double center = argc > 4 ? std::atof(argv[4]) : 160e3;
            const double cutoff = argc > 5 ? std::atof(argv[5]) : 0.0;
            auto src = mode == "raw" ? gr::amps::recc_fused::make(10, 400e3, center, 2, cutoff) : gr::amps::recc_fused::make(10);
            if (mode == "iqb") gr::msg_connect(src, "bursts", dec, "bursts"); else gr::msg_connect(src, "records", dec, "records");
            const size_t ns = data.size() / 8;
            for (size_t off = 0; off < ns; off += (size_t)chunk) {
                int n = (int)std::min<size_t>((size_t)chunk, ns - off);
                gr_vector_const_void_star ins = { data.data() + 8 * off };
                if (src->work(n, ins, outs) != 0) return 1;
            }
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    return 0;
}
