# Makefile -- the in-tree build without Python: the gfx950 C-ABI library, the host C++ blocks, the CPU checker and the examples.
# (`python -c "import __graft_entry__ as g; g.build()"` runs the same commands; gr_amps_amd/build.py and gr_amps_amd/host/__init__.py
# hold them for the test suite.)  hipcc cross-compiles gfx950 without a GPU; nothing here needs one.
#
#   make            library + host blocks + recctest + oracle
#   make examples   examples/recc_abi_example (plain C99 against include/amps_recc.h)
#   make check      the CPU test suite;  make check-gpu  the parity suite on an MI355X
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CC      ?= gcc
PKG      = gr_amps_amd
CSRC     = $(PKG)/csrc
HOST     = $(PKG)/host
# -ffp-contract=off: the float stage is specified operation by operation (include/amps_recc_numerics.h);
# -fno-slp-vectorize: SLP packs the demod into v_pk_* + v_mov shuffles: -10 % (measured)
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function
HOSTINC  = -I$(HOST)/include -I$(HOST)/gr_min -Iinclude
HOSTSRC  = $(HOST)/lib/recc_impl.cc $(HOST)/lib/recc_decode_impl.cc $(HOST)/lib/recc_fused_impl.cc $(HOST)/lib/recc_bank_impl.cc $(HOST)/lib/recc_wideband_impl.cc

all: $(PKG)/libamps_recc.so $(PKG)/libgnuradio-amps-mi355x.so $(PKG)/recctest oracle/libamps_oracle.so

$(PKG)/libamps_recc.so: $(CSRC)/amps_recc.hip $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)
	$(HIPCC) $(HIPFLAGS) -Iinclude -I$(CSRC) $(CSRC)/amps_recc.hip -o $@

$(PKG)/libgnuradio-amps-mi355x.so: $(HOSTSRC) $(wildcard $(HOST)/lib/*.h) $(HOST)/gr_min/gnuradio_min.h include/amps_recc.h $(PKG)/libamps_recc.so
	$(CXX) -std=c++17 -O2 -fPIC -Wall $(HOSTINC) -shared -o $@ $(HOSTSRC) -L$(PKG) -lamps_recc -Wl,-rpath,'$$ORIGIN'

$(PKG)/recctest: $(HOST)/apps/recctest.cc $(PKG)/libgnuradio-amps-mi355x.so
	$(CXX) -std=c++17 -O2 -fPIC -Wall $(HOSTINC) -o $@ $< -L$(PKG) -lgnuradio-amps-mi355x -lamps_recc -Wl,-rpath,'$$ORIGIN'

oracle/libamps_oracle.so: oracle/ref_chain.c oracle/fused_model.c oracle/amps_oracle.h $(wildcard include/*.h)
	$(MAKE) -C oracle

examples: examples/recc_abi_example
examples/recc_abi_example: examples/recc_abi_example.c include/amps_recc.h $(PKG)/libamps_recc.so
	$(CC) -std=c99 -Wall -Wextra -pedantic -Iinclude $< -o $@ -L$(PKG) -lamps_recc -Wl,-rpath,$(CURDIR)/$(PKG) -Wl,-rpath-link,/opt/rocm/lib

check: all
	python -m pytest tests -x -q -m "not gpu"

clean:
	rm -f $(PKG)/libamps_recc.so $(PKG)/libgnuradio-amps-mi355x.so $(PKG)/recctest examples/recc_abi_example
	$(MAKE) -C oracle clean

.PHONY: all examples check check-gpu clean

check-gpu: all
	python -m pytest tests -x -q -m gpu
