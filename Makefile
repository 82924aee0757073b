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

# The resident tracker kernels run at the 256-register limit of a 512-thread block.  LLVM's machine LICM hoists the
# materialisation of every literal (fp64 polynomial coefficients, addresses) out of the Gauss-Newton iteration loop and
# then spills those registers to scratch (up to 216 B per lane, reloaded inside the pixel passes and the scalar solve);
# without it the same kernels need 170-255 registers and no scratch.
$(CSRC)/track.o: HIPFLAGS += -mllvm -disable-machine-licm

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
