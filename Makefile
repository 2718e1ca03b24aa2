# Build everything in-tree (no cmake / no node-gyp): the HIP C-ABI library, the N-API addon,
# the test oracle and the synthetic-text generator.  `python __graft_entry__.py` calls this.
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CC        ?= gcc
CXX       ?= g++
PKG       := compressjs-flattened_amd
CSRC      := $(PKG)/csrc
NODE_INC  ?= /usr/include/node

HIPFLAGS  := $(XFLAGS) -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
HIP_SRCS  := $(wildcard $(CSRC)/*.hip)
HIP_HDRS  := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)
HIP_OBJS  := $(HIP_SRCS:.hip=.o)

all: hip oracle textgen napi

hip: $(PKG)/libcjs_hip.so
$(CSRC)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(PKG)/libcjs_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS) -lpthread

oracle: oracle/libcjs_oracle.so
oracle/libcjs_oracle.so: oracle/cjs_oracle.c oracle/cjs_oracle.h
	$(CC) -O2 -std=c99 -Wall -fPIC -shared -o $@ oracle/cjs_oracle.c

textgen: tools/libcjs_textgen.so tools/textgen
tools/libcjs_textgen.so: tools/textgen.c
	$(CC) -O2 -std=c99 -Wall -fPIC -shared -o $@ tools/textgen.c
tools/textgen: tools/textgen.c
	$(CC) -O2 -std=c99 -Wall -DTEXTGEN_MAIN -o $@ tools/textgen.c

# N-API addon: thin shim over the C ABI (dlopen()s libcjs_hip.so next to it at load time)
ifneq ($(wildcard $(PKG)/js/cjs_napi.cc),)
napi: $(PKG)/js/cjs_napi.node
$(PKG)/js/cjs_napi.node: $(PKG)/js/cjs_napi.cc include/cjs_hip.h
	@if [ -f $(NODE_INC)/node_api.h ]; then \
	  $(CXX) -O2 -std=c++17 -fPIC -shared -I$(NODE_INC) -Iinclude -o $@ $(PKG)/js/cjs_napi.cc -ldl ; \
	else echo "node_api.h not found: skipping N-API addon"; fi
else
napi:
	@echo "N-API addon source not present yet"
endif

clean:
	rm -f $(CSRC)/*.o $(PKG)/libcjs_hip.so oracle/libcjs_oracle.so tools/libcjs_textgen.so tools/textgen $(PKG)/js/cjs_napi.node

.PHONY: all hip oracle textgen napi clean
