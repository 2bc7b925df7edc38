# A/B and timing-ablation builds of the library (NOT product code).  Run from sepreformer_amd/csrc:
#   make -f ../../tools/variants.mk ../_native/libsepr_hip_<tag>.so      then      SEPR_LIB_VARIANT=<tag> python ...
# Each variant is a one-shot whole-library build with extra -D switches (see the SEPR_* macros at the top of each .hip file).
include Makefile
# projection core: persistence / stagger, load / store / weight / MFMA ablations, LDS pads
VARIANT_np = -DSEPR_GEMM_PERSIST=0 -DSEPR_GEMM_STAGGER=0
VARIANT_ns = -DSEPR_GEMM_STAGGER=0
VARIANT_ablst = -DSEPR_ABL_NOSTORE=1
VARIANT_ablld = -DSEPR_ABL_NOLOAD=1
VARIANT_ablw = -DSEPR_ABL_NOW=1
VARIANT_ablmma = -DSEPR_ABL_NOMMA=1
VARIANT_ablall = -DSEPR_ABL_NOLOAD=1 -DSEPR_ABL_NOSTORE=1 -DSEPR_ABL_NOW=1
VARIANT_x3p8 = -DSEPR_X3_LDK_PAD=8
VARIANT_gbp8 = -DSEPR_GB_LDK_PAD=8
# fused GCFN: 6 x 14-frame waves everywhere
VARIANT_gfmt1 = -DSEPR_GF3_MT=1
# EGA attention timing ablations (bit mask, profiles/r02_v5_attention_ablation.txt)
VARIANT_at1 = -DSEPR_AT_ABL=1
VARIANT_at2 = -DSEPR_AT_ABL=2
VARIANT_at4 = -DSEPR_AT_ABL=4
VARIANT_at8 = -DSEPR_AT_ABL=8
VARIANT_at16 = -DSEPR_AT_ABL=16
VARIANT_at31 = -DSEPR_AT_ABL=31
# weight-gradient contraction ablations (profiles/r03_v3_gemm_tn_ablation.txt)
VARIANT_tnmap = -DSEPR_TN_LANEMAP=1
VARIANT_tn1 = -DSEPR_TN_ABL=1
VARIANT_tn2 = -DSEPR_TN_ABL=2
VARIANT_tn3 = -DSEPR_TN_ABL=3
VARIANT_tn4 = -DSEPR_TN_ABL=4
VARIANT_tn7 = -DSEPR_TN_ABL=7
ablations: $(OUTDIR)/libsepr_hip_ablst.so $(OUTDIR)/libsepr_hip_ablld.so $(OUTDIR)/libsepr_hip_ablw.so $(OUTDIR)/libsepr_hip_ablmma.so $(OUTDIR)/libsepr_hip_ablall.so
$(OUTDIR)/libsepr_hip_%.so: $(SRCS) $(HDRS)
	@mkdir -p $(OUTDIR)
	$(HIPCC) $(CXXFLAGS) $(VARIANT_$*) -shared $(SRCS) -o $@
	@$(PYTHON) ../../tools/isa_lint.py $@ || echo "WARNING: $@ contains the gfx950-faulty packed-f32 form (DESIGN.md section 10): its results may differ run to run - timing only"
VARIANT_resx = -DSEPR_GF3_RESX=1
# conv-fold proxies of the fused GCFN (round-4 review item 6; wrong results, timing only)
VARIANT_gfold32 = -DSEPR_GF_ABL=32
VARIANT_gfold64 = -DSEPR_GF_ABL=64
# table-gradient kernel of the EGA attention backward: 16-row load batches (round 4 A/B)
VARIANT_band16 = -DSEPR_AXB_ROWS=16 -DSEPR_AXB_BPC=2
VARIANT_bandg2 = -DSEPR_AXB_GROUP=2
VARIANT_bandg1 = -DSEPR_AXB_GROUP=1
VARIANT_noslp = -fno-slp-vectorize
# Base inference attention: 8 waves (128 queries) per workgroup (round 5 A/B)
VARIANT_atnw8 = -DSEPR_AT_NW=8
# plain-bf16 weight-gradient contraction compiled for 3 waves per SIMD (round 5 A/B; pair with SEPR_TN_WGS=768)
VARIANT_tn3w = -DSEPR_TN_ONE_WPE=3
# round 6: timing ablations of the latency-form fused GCFN (batch 1): no chunk loop / no copies after the ring prologue / one fragment read per chunk / no exp+rcp
VARIANT_gfa1 = -DSEPR_GF_ABL=1
VARIANT_gfa2 = -DSEPR_GF_ABL=2
VARIANT_gfa8 = -DSEPR_GF_ABL=8
VARIANT_gfa16 = -DSEPR_GF_ABL=16
VARIANT_gfa10 = -DSEPR_GF_ABL=10
# round 6: inline-asm LDS-DMA in the (non-latency) fused GCFN kernels
VARIANT_gfasm = -DSEPR_GF3_ASMDMA=1
VARIANT_noasm = -DSEPR_GF3_ASMDMA=0 -DSEPR_CF_ASMDMA=0 -DSEPR_SPK_ASMDMA=0
VARIANT_noasmcf = -DSEPR_CF_ASMDMA=0 -DSEPR_SPK_ASMDMA=0
# round 6, review item 4: what does the epilogue's second read of x cost?  gfa128 = no residual read (wrong results), gfa256 = residual read from a cold range
VARIANT_gfa128 = -DSEPR_GF_ABL=128
VARIANT_gfa256 = -DSEPR_GF_ABL=256
# round 6: LDS fragment read-ahead / phase order of the fused GCFN, measured on the F = 256 (one-wave-per-SIMD) instantiation
VARIANT_gfr3 = -DSEPR_GF3_RING=3
VARIANT_gfr4 = -DSEPR_GF3_RING=4
VARIANT_gfuf0 = -DSEPR_GF3_UPFIRST=0
# round 6: wait states between the SGPR-base set-up and the inline-asm LDS-DMA (product: 4 = hazard-safe; 0 = the first form, A/B only)
VARIANT_gldsnop0 = -DSEPR_GLDS_NOP=0
VARIANT_gfnofence = -DSEPR_GF3_FENCE256=0
# round 6: timing ablations of the hidden-split GCFN form (wrong results): 512 = weight fragments requested once, 1024 = no conv / GLU, 1536 = both
VARIANT_hsa512 = -DSEPR_GF_ABL=512
VARIANT_hsa1024 = -DSEPR_GF_ABL=1024
VARIANT_hsa1536 = -DSEPR_GF_ABL=1536
# round 6, second session: row-window epilogue of gcfn_bwd_mid_kernel - gbepi0 = through LDS tiles (the rounds 2-5 form), gbepi2w / gbepi3w = in registers with the
# plane-staged kernel compiled and launched for two / three workgroups per CU (profiles/r06_gcfn_bwd_regepi.txt)
VARIANT_gbepi0 = -DSEPR_GB_REGEPI=0 -DSEPR_GB_PL_WGS=3
VARIANT_gbepi2w = -DSEPR_GB_REGEPI=1 -DSEPR_GB_PL_WGS=2
VARIANT_gbepi3w = -DSEPR_GB_REGEPI=1 -DSEPR_GB_PL_WGS=3
# round 6, second session: bf16-source projections of the plain-bf16 step (GCFN input gradient) with the widened one-slab staging of rounds 4-5 (A/B of the raw two-slab form)
VARIANT_x3raw0 = -DSEPR_X3_RAW16=0
VARIANT_x3deep0 = -DSEPR_X3_DEEP16=0
# round 6, last session: plane-staged GCFN backward middle kernel (profiles/r06_gcfn_bwd_waits.txt).  Switches (sepr_gcfn_bwd_fused.hip): SEPR_GB_TOPWAIT = a
# compiler-visible vmcnt(0) at the top of a tile (hipcc otherwise puts its own vmcnt(0) BETWEEN the tile's copies, tools/isa_trace.py), SEPR_GB_ONEBAR = one wait +
# barrier for all of a tile's slabs, SEPR_GB_CONSTLDS = the column block's depthwise taps / biases parked in LDS once per persistent workgroup (product: 1),
# SEPR_GB_SLIDE = conv windows shared between a thread's rows (bit 1 pass A, bit 2 pass B; product: 2).  Every variant spells its switches out (the product defaults moved during the session).
GB0 = -DSEPR_GB_TOPWAIT=0 -DSEPR_GB_ONEBAR=0 -DSEPR_GB_CONSTLDS=0 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
GBC = -DSEPR_GB_TOPWAIT=0 -DSEPR_GB_ONEBAR=0 -DSEPR_GB_CONSTLDS=1
VARIANT_gbcs0 = $(GB0)
VARIANT_gbonebar = -DSEPR_GB_TOPWAIT=0 -DSEPR_GB_ONEBAR=1 -DSEPR_GB_CONSTLDS=0 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbtopwait = -DSEPR_GB_TOPWAIT=1 -DSEPR_GB_ONEBAR=0 -DSEPR_GB_CONSTLDS=0 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbtopone = -DSEPR_GB_TOPWAIT=1 -DSEPR_GB_ONEBAR=1 -DSEPR_GB_CONSTLDS=0 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbconst = -DSEPR_GB_TOPWAIT=1 -DSEPR_GB_ONEBAR=0 -DSEPR_GB_CONSTLDS=1 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbconstone = -DSEPR_GB_TOPWAIT=1 -DSEPR_GB_ONEBAR=1 -DSEPR_GB_CONSTLDS=1 -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbcs = $(GBC) -DSEPR_GB_SLIDE=0 -DSEPR_GB_REDERIVE=0
VARIANT_gbcss = $(GBC) -DSEPR_GB_SLIDE=3 -DSEPR_GB_REDERIVE=0
VARIANT_gbslide = -DSEPR_GB_TOPWAIT=1 -DSEPR_GB_ONEBAR=0 -DSEPR_GB_CONSTLDS=1 -DSEPR_GB_SLIDE=3 -DSEPR_GB_REDERIVE=0
VARIANT_gbcsb = $(GBC) -DSEPR_GB_SLIDE=2 -DSEPR_GB_REDERIVE=0
# SEPR_GB_REDERIVE = the thread index made opaque at the top of every tile, so the per-thread LDS addresses / lane roles derived from it are recomputed per tile instead of
# being hoisted out of the tile loop and kept alive (or spilled) across all phases: 168 registers + 10 spilled -> 157, none spilled (bit 1: the plane-staged form, bit 2: the
# register-staged forms; product: 3)
VARIANT_gbred = $(GBC) -DSEPR_GB_SLIDE=2 -DSEPR_GB_REDERIVE=1 -DSEPR_GF3_REDERIVE=0 -DSEPR_XW_REDERIVE=0
VARIANT_gbred3 = $(GBC) -DSEPR_GB_SLIDE=3 -DSEPR_GB_REDERIVE=1 -DSEPR_GF3_REDERIVE=0 -DSEPR_XW_REDERIVE=0
VARIANT_gbredepi = $(GBC) -DSEPR_GB_SLIDE=2 -DSEPR_GB_REDERIVE=1 -DSEPR_GB_REGEPI=1 -DSEPR_GB_PL_WGS=3 -DSEPR_GF3_REDERIVE=0 -DSEPR_XW_REDERIVE=0
VARIANT_gbredall = $(GBC) -DSEPR_GB_SLIDE=2 -DSEPR_GB_REDERIVE=3 -DSEPR_GF3_REDERIVE=0
VARIANT_x3train = $(GBC) -DSEPR_GB_SLIDE=2 -DSEPR_GB_REDERIVE=3 -DSEPR_GF3_REDERIVE=1
# round 6, last session: fused GCFN forward with the thread index made opaque per tile (1), in front of the epilogue (2), both (3): 256 registers + 7 spilled -> 245 / 241, none spilled
VARIANT_gfred0 = -DSEPR_GF3_REDERIVE=0
VARIANT_gfred1 = -DSEPR_GF3_REDERIVE=1
VARIANT_gfred2 = -DSEPR_GF3_REDERIVE=2
VARIANT_gfred3 = -DSEPR_GF3_REDERIVE=3
# round 6, last session: wide projection core (Large) with the thread index made opaque per tile: every instantiation spill-free except <1,7,1> (20 -> 3 spilled registers)
VARIANT_xwred0 = -DSEPR_XW_REDERIVE=0
VARIANT_xwred = -DSEPR_XW_REDERIVE=1
