# Builds the engine without Python (the same recipe __graft_entry__.build() runs).
HIPCC   ?= hipcc
CC      ?= gcc
LIBDIR  := hpg-variant_amd/lib
CSRC    := hpg-variant_amd/csrc
HOST    := hpg-variant_amd/host

all: $(LIBDIR)/libhpgv.so $(LIBDIR)/libhpgv_host.so oracle

$(LIBDIR)/libhpgv.so: $(CSRC)/hpgv_capi.hip $(wildcard $(CSRC)/*.h) include/hpgv.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wextra -o $@ $(CSRC)/hpgv_capi.hip

$(LIBDIR)/libhpgv_host.so: $(HOST)/hpgv_host.c include/hpgv_host.h include/hpgv.h $(LIBDIR)/libhpgv.so
	$(CC) -O2 -g -std=gnu99 -fPIC -shared -fopenmp -Wall -Wextra -Iinclude -o $@ $(HOST)/hpgv_host.c \
	    -L$(LIBDIR) -lhpgv -Wl,-rpath,'$$ORIGIN' -lm -lz

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIBDIR)/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
