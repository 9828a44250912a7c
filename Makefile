# Builds the C-ABI HIP library for gfx950 (cross-compiles without a GPU) and the oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := mpnn_amd/csrc
OUT   := mpnn_amd/libmpnn_hip.so
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wno-unused-result $(EXTRA)
OBJS  := $(CSRC)/mpnn_featurizer.o $(CSRC)/mpnn_edge.o $(CSRC)/mpnn_edge_bwd.o $(CSRC)/mpnn_node.o $(CSRC)/mpnn_node_bwd.o $(CSRC)/mpnn_data.o $(CSRC)/mpnn_loss.o $(CSRC)/mpnn_abi.o \
         $(CSRC)/mpnn_featurizer_f16.o $(CSRC)/mpnn_edge_f16.o $(CSRC)/mpnn_node_f16.o

all: $(OUT)

# neighbour search must round like the reference's separate ATen ops: no FMA contraction there
# -fno-slp-vectorize: packed fp32 VALU forms occupy the issue port ~10 cycles next to MFMAs against 4.2 for the scalar
# form (tools/ubench/coissue.hip); measured 0.318 -> 0.303 ms on mpnn_edge_features
$(CSRC)/mpnn_featurizer.o: $(CSRC)/mpnn_featurizer.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -fno-slp-vectorize -c $< -o $@

$(CSRC)/mpnn_edge.o: $(CSRC)/mpnn_edge.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h $(CSRC)/mpnn_sample.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/mpnn_edge_bwd.o: $(CSRC)/mpnn_edge_bwd.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h $(CSRC)/mpnn_bwd_common.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/mpnn_abi.o: $(CSRC)/mpnn_abi.cpp $(CSRC)/mpnn_ablation.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/mpnn_data.o: $(CSRC)/mpnn_data.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/mpnn_loss.o: $(CSRC)/mpnn_loss.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(OUT)

# Every compile-time switch left in the sources, compiled once (front end + device code generation, nothing linked):
# the f16-product twin is part of `all`; here the phase stamps and each timing-only ablation of mpnn_ablation.h.
ABL_EDGE := NOXLOAD XSAME ROWSAME W3LDS NORES NOLN NOSTORE LEVEL_NOFILL LEVEL_NOMSG LEVEL_NONODE NOGELU NOREDUCE NOMFMA \
            WSTREAM_SAME WSTREAM_NONE NOROW SAMPLE_NOGEMM SAMPLE_NOLOOP
ABL_NODE := NODMA SAMPLE_NOTAIL NOGELU NOMFMA
ABL_BWD  := WGRAD_STREAM
check-switches:
	@set -e; for f in $(ABL_EDGE); do echo "mpnn_edge.hip -DMPNN_ABL_$$f"; \
	  $(HIPCC) $(HIPFLAGS) -DMPNN_ABL_$$f -c $(CSRC)/mpnn_edge.hip -o /dev/null; done
	@set -e; for f in $(ABL_NODE); do echo "mpnn_node.hip -DMPNN_ABL_$$f"; \
	  $(HIPCC) $(HIPFLAGS) -DMPNN_ABL_$$f -c $(CSRC)/mpnn_node.hip -o /dev/null; done
	@set -e; for f in $(ABL_BWD); do echo "mpnn_edge_bwd.hip -DMPNN_ABL_$$f"; \
	  $(HIPCC) $(HIPFLAGS) -DMPNN_ABL_$$f -c $(CSRC)/mpnn_edge_bwd.hip -o /dev/null; done
	@echo "mpnn_edge.hip -DMPNN_STAMP"; $(HIPCC) $(HIPFLAGS) -DMPNN_STAMP -c $(CSRC)/mpnn_edge.hip -o /dev/null
	@echo "check-switches: all switch builds compile"

# Wait states around the inline-asm VALU instructions (hipcc pads none for them) and the operand each v_fma_mix reads:
# tools/hazard_scan.py over the final assembly of every source that holds asm, in both precisions.
check-hazards:
	@set -e; for f in edge node edge_bwd node_bwd loss; do python3 tools/hazard_scan.py $(CSRC)/mpnn_$$f.hip | tail -1; done
	@python3 tools/hazard_scan.py $(CSRC)/mpnn_featurizer.hip -ffp-contract=off -fno-slp-vectorize | tail -1
	@set -e; for f in edge node; do python3 tools/hazard_scan.py $(CSRC)/mpnn_$$f.hip -DMPNN_FAST | tail -1; done
	@python3 tools/hazard_scan.py $(CSRC)/mpnn_featurizer.hip -DMPNN_FAST -ffp-contract=off -fno-slp-vectorize | tail -1
	@echo "check-hazards: no findings"

.PHONY: all clean check-switches check-hazards

$(CSRC)/mpnn_node.o: $(CSRC)/mpnn_node.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_sample.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/mpnn_node_bwd.o: $(CSRC)/mpnn_node_bwd.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h $(CSRC)/mpnn_bwd_common.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# second compilation of the forward kernels with one f16 product per fp32 product (entry points *_f16)
$(CSRC)/mpnn_featurizer_f16.o: $(CSRC)/mpnn_featurizer.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h
	$(HIPCC) $(HIPFLAGS) -DMPNN_FAST -ffp-contract=off -fno-slp-vectorize -c $< -o $@

$(CSRC)/mpnn_edge_f16.o: $(CSRC)/mpnn_edge.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h $(CSRC)/mpnn_edge_common.h $(CSRC)/mpnn_sample.h
	$(HIPCC) $(HIPFLAGS) -DMPNN_FAST -c $< -o $@

$(CSRC)/mpnn_node_f16.o: $(CSRC)/mpnn_node.hip $(CSRC)/mpnn_common.h $(CSRC)/mpnn_ablation.h
	$(HIPCC) $(HIPFLAGS) -DMPNN_FAST -c $< -o $@
