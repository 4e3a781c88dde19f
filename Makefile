# Builds the engine without Python (the same recipe __graft_entry__.build() runs).
HIPCC   ?= hipcc
CC      ?= gcc
LIBDIR  := hpg-variant_amd/lib
CSRC    := hpg-variant_amd/csrc
HOST    := hpg-variant_amd/host

all: $(LIBDIR)/libhpgv.so $(LIBDIR)/libhpgv_host.so oracle

HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-function
UNITS   := $(wildcard $(CSRC)/*.hip)
OBJS    := $(patsubst $(CSRC)/%.hip,$(LIBDIR)/obj/%.o,$(UNITS))

# the bgzip decoder's wave-per-block kernel branches on wave-uniform values only: keep its control flow as written
$(LIBDIR)/obj/hpgv_inflate_capi.o: UNITFLAGS := -mllvm -structurizecfg-skip-uniform-regions

$(LIBDIR)/obj/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/hpgv.h
	@mkdir -p $(LIBDIR)/obj
	$(HIPCC) $(HIPFLAGS) $(UNITFLAGS) -c -o $@ $<

$(LIBDIR)/libhpgv.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

$(LIBDIR)/libhpgv_host.so: $(wildcard $(HOST)/*.c) $(HOST)/hpgv_host_internal.h include/hpgv_host.h include/hpgv.h $(LIBDIR)/libhpgv.so
	$(CC) -O2 -g -std=gnu99 -fPIC -shared -fopenmp -Wall -Wextra -Iinclude -I$(HOST) -o $@ $(wildcard $(HOST)/*.c) \
	    -L$(LIBDIR) -lhpgv -Wl,-rpath,'$$ORIGIN' -lm -lz

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIBDIR)/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
