"""A second restatement of gr::amps::recc_impl::work (lib/recc_impl.cc:93-145), written apart from oracle/ref_chain.c --
TEST INFRASTRUCTURE ONLY.  Python over a bytearray; `bytes.find` stands where the reference calls memmem.

State (lib/recc_impl.h:31-43): the 64 KiB symbol buffer, its fill level, and the pending trigger position (None = NULL).
work() is transcribed statement by statement; the comments give the reference line of each."""

BUFSZ, WINDOW, CAPTURE = 65536, 4096, 3374
_TRIG_BITS = "1010101010101010101010101011100010010"                      # lib/recc_impl.cc:76
TRIGGER = bytes(b for ch in _TRIG_BITS for b in ((0, 1) if ch == "1" else (1, 0)))   # :51-65  '0' -> (1,0), '1' -> (0,1)


class ReccWork2:
    def __init__(self):
        self.buf = bytearray(BUFSZ)
        self.len = 0
        self.cur = None

    def work(self, chunk):
        """one work() call with noutput_items = len(chunk); returns the published 3374-byte payload or None"""
        n = len(chunk)
        if n < 1:                                                         # :99-102
            return None
        assert n < BUFSZ - WINDOW                                         # :103 (the caller's contract)
        if self.len + n > BUFSZ:                                          # :104-108  wrap: the LAST 4096 bytes OF THE BUFFER, pending trigger forgotten
            self.buf[0:WINDOW] = self.buf[BUFSZ - WINDOW:BUFSZ]
            self.len = WINDOW
            self.cur = None
        self.buf[self.len:self.len + n] = bytes(chunk)                    # :110
        self.len += n                                                     # :111
        out = None
        if self.len > len(TRIGGER):                                       # :114
            search = min(self.len, n + len(TRIGGER) - 1)                  # :115
            if self.cur is None:                                          # :117-119
                p = bytes(self.buf[self.len - search:self.len]).find(TRIGGER)
                self.cur = None if p < 0 else self.len - search + p
            if self.cur is not None:                                      # :121
                start = self.cur                                          # :122
                captured = self.len - start - len(TRIGGER)                # :124
                if captured > CAPTURE:                                    # :125  strictly more
                    out = bytes(self.buf[start + len(TRIGGER):start + len(TRIGGER) + CAPTURE])   # :126
                    tomove = self.len - (captured + len(TRIGGER))         # :129  (= start)
                    if tomove > 0:                                        # :131-133  memmove of possibly overlapping ranges: copy first
                        src = captured + len(TRIGGER)
                        self.buf[0:tomove] = bytes(self.buf[src:src + tomove])
                    self.len -= tomove                                    # :134  (sic)
                    self.cur = None                                       # :135
        return out

    def run(self, stream, schedule):
        stream = bytes(stream)
        out, off, call = [], 0, 0
        while off < len(stream):
            n = min(int(schedule[call % len(schedule)]), len(stream) - off)
            b = self.work(stream[off:off + n])
            if b is not None:
                out.append((call, b))
            off += n
            call += 1
        return out
