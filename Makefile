# Build of the MI355X (gfx950) library and the CPU oracle.  `make` = both.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := densemonoslam_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude
LIB   := densemonoslam_amd/libdmslam_hip.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(SRCS:.hip=.o)
HDRS  := $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
