/*
 * oriana_hip.h -- C ABI of the MI355X (gfx950) CAVI engine for probabilistic count matrix
 * factorisation.  This is the drop-in boundary for the hot path of AntoinePassemiers/Oriana:
 * every entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  Plain pointers and sizes only; all pointers are DEVICE pointers unless a
 * parameter is documented as host.  Calls are asynchronous on `stream` (a hipStream_t passed as
 * void*; NULL = the default stream).  Return value: 0 on success, a negative code otherwise
 * (-1000 - hipError_t for HIP errors, ORIANA_E* for argument errors); nothing is thrown.
 * The library keeps no global state; all scratch memory is supplied by the caller.
 *
 * Layouts
 *   - "dense (r, K) f32/f64": C-contiguous, exactly the reference's ndarray layout
 *     (oriana/parameters.py:8-32 holds float64 buffers; the kernels take float32, gap.py:67).
 *   - "factor matrix (r, Kp) f32": K padded with zeros to Kp = oriana_kpad(K); row-major.
 *   - oriana_counts: the count matrix X (constant across sweeps, oriana/models/gap.py:29-32)
 *     repacked once into 256 x 256 tiles that keep only the non-zero counts; see DESIGN.md.
 */
#ifndef ORIANA_HIP_H
#define ORIANA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORIANA_TILE 256

#define ORIANA_EINVAL   (-1)   /* bad argument (NULL pointer, negative size, K out of range) */
#define ORIANA_EKRANGE  (-2)   /* K larger than the largest compiled configuration */
#define ORIANA_EQUIRK   (-3)   /* reference_quirks needs K <= m (zigap.py:94 reads D_hat[i, k]) */
#define ORIANA_EUNIT    (-4)   /* a hybrid resident handle serves the ZI nests under oriana_counts_declare_unit_dropout only */

/* 8-byte record of one slot of the row-side stream.  x == 0 marks a padding slot. */
typedef struct {
    float    x;     /* the count, as float32 (gap.py:94 casts X to float32); 0 = padding */
    uint16_t cdst;  /* slot of this entry inside the tile's column-side region */
    uint8_t  col;   /* column inside the tile */
    uint8_t  pad;
} oriana_rowrec;

/* Tiled, sliced non-zero layout of one row shard of X (built once by oriana_pack_*).
 *
 * X is cut into 256 x 256 tiles (row-block-major).  Inside a tile the non-zeros are stored twice:
 *   - row side: the tile's 256 rows form 16 slices of 16 rows.  A slice is a sequence of
 *     "iterations" of 64 slots = 16 rows x 4 consecutive records of that row (slot = iteration*64 +
 *     row_in_slice*4 + u); a row's records are in increasing column order; rows shorter than the
 *     longest row of the slice are padded (x = 0).  One wave-wide 512-byte load fetches an iteration.
 *   - column side: the same with 16-column slices; slot = iteration*64 + col_in_slice*4 + u holds
 *     the tile row index (ridx) and, in the per-sweep array `s`, the scalar s_ij of that entry;
 *     a column's entries are in increasing row order.  The last 64 slots of a tile's column-side
 *     region are write-only dummies (target of the padding lanes' stores).
 * All arrays live in device memory and are owned by the caller (the Python host allocates them as
 * torch tensors; rowrec and ridx must be zero-filled before oriana_pack_fill). */
typedef struct {
    int64_t n, m;                 /* rows (cells) in this shard, columns (genes) */
    int64_t nrb, ncb;             /* ceil(n / 256), ceil(m / 256) */
    int64_t nnz;                  /* non-zero entries */
    int64_t rslots, cslots;       /* total slots of the row-side / column-side streams */
    const int64_t       *roff;       /* [nrb*ncb + 1] first row-side slot of tile (rb, cb), rb-major */
    const int64_t       *coff;       /* [nrb*ncb + 1] first column-side slot of the tile */
    const uint32_t      *rslice;     /* [nrb*ncb][17] slot offsets of the 16-row slices inside the tile */
    const uint32_t      *cslice;     /* [nrb*ncb][17] slot offsets of the 16-column slices inside the tile */
    const oriana_rowrec *rowrec;     /* [rslots] */
    const uint8_t       *ridx;       /* [cslots] row inside the tile */
    /* Optional internal orderings (NULL = identity): packed column c holds gene col_perm[c], packed
     * row r holds cell row_perm[r].  Sorting genes (and cells) by their non-zero count puts
     * entries of similar density in the same tile, which shortens the padding of the slices.
     * All dense inputs / outputs of the API stay in the caller's order: the permutation is applied
     * by oriana_factor_prep (gather), oriana_finalize (scatter) and oriana_fixup. */
    const int32_t       *col_perm;   /* [m] */
    const int32_t       *row_perm;   /* [n] */
} oriana_counts;

/* Kp for a given K (0 if K is out of range). */
int64_t oriana_kpad(int64_t K);
/* Column tiles (of 256 genes) one work-group of oriana_col_pass covers for this K: the "column block" of a work
 * item indexes groups of this many adjacent tiles (2 for K <= 116, where one image of the row block serves two
 * tiles; 1 for the wider K; 0 if K is out of range). */
int64_t oriana_col_block_tiles(int64_t K);
/* Library / build identification ("oriana_hip gfx950 <version>"). */
const char *oriana_version(void);

/* ---- packing the count matrix (replaces `self.X[:].astype(np.float32)`, gap.py:94) --------
 * Two passes over dense row chunks (a chunk starts at a multiple of 256 rows):
 *   1. oriana_pack_count   -> per-tile nnz / slot counts and the slice offsets
 *   2. (caller) exclusive scans of tile_rslots, tile_cslots -> roff, coff; zero-filled rowrec, ridx
 *   3. oriana_pack_fill    -> rowrec / ridx (/ side_nz)
 * X is dense (rows, m) with leading dimension ldx (elements), float32 or (xdtype = 1) int64 /
 * (xdtype = 2) int32 / (xdtype = 3) float64.  rb0 = first row block of the chunk.
 */
int oriana_pack_count(const void *X, int xdtype, int64_t rows, int64_t m, int64_t ldx,
                      int64_t rb0, int64_t ncb,
                      int32_t *tile_nnz,      /* [nrb*ncb] */
                      int32_t *tile_rslots,   /* [nrb*ncb] */
                      int32_t *tile_cslots,   /* [nrb*ncb] (includes the 64 dummy slots) */
                      uint32_t *rslice,       /* [nrb*ncb][17] */
                      uint32_t *cslice,       /* [nrb*ncb][17] */
                      void *stream);
int oriana_pack_fill(const void *X, int xdtype, int64_t rows, int64_t m, int64_t ldx,
                     int64_t rb0, int64_t ncb, const int64_t *roff, const int64_t *coff,
                     const uint32_t *rslice, const uint32_t *cslice,
                     oriana_rowrec *rowrec, uint8_t *ridx,
                     /* optional: gather a dense (rows, m) f32 side matrix (D_hat) at the non-zeros,
                      * row-side slot order */
                     const float *side, int64_t ldside, float *side_nz,
                     void *stream);

/* ---- factor preparation ---------------------------------------------------------------------
 * From E[log U] (gamma.py:52-61 output, dense (r, K) f32) build the factor matrix
 *   F[i, k] = exp(l[i, k] - mu[i]) * (mask ? mask[i, k] : 1),   mu[i] = max_k l[i, k],
 * the shift being undone analytically (softmax is shift invariant, gap.py:74-78).  Rows whose
 * shift is too large for the shifted form to reproduce the reference's float32 behaviour are
 * zero-filled: every entry that touches them fails the den test and is evaluated by the exact slow
 * path (oriana_fixup).
 */
int oriana_factor_prep(float *F, float *mu, const float *logF, const float *mask,
                       const int32_t *row_index,   /* F row i is built from logF row row_index[i] (NULL: i) */
                       int64_t r, int64_t K, void *stream);
/* Both sides at once, with the validity test CENTRED on the two sides' typical shifts: only the sums
 * lu[i,k] + lv[j,k] enter the loop nests (gap.py:74), and the iterations drift along the scale indeterminacy
 * (U c, V / c) -- a row takes the shifted form iff its shift lies within A_side of the mean shift of its side, A_u + A_v
 * chosen (and shared in proportion to the sides' spreads) so that every sum of two accepted shifts keeps the
 * reference's own float32 denominator normal.  Same F as
 * oriana_factor_prep wherever both accept a row.  maskV: S_tilde of the sparse models or NULL; scratch:
 * oriana_prep_scratch_bytes() bytes of device memory, ZEROED once before the first call (per-group partial sums of the
 * row maxima, added up in a fixed order by the last group to finish: no float atomics, nothing to clear between
 * calls, safe inside a captured graph). */
int oriana_factor_prep_pair(float *FU, float *FV, const float *logU, const float *logV, const float *maskV,
                            const int32_t *row_index_u, const int32_t *row_index_v,
                            int64_t n, int64_t m, int64_t K, float *scratch, void *stream);
/* The same with a list of buffers that the second launch zero-fills on the side (the outputs and scratch the passes of
 * a sweep accumulate into -- Z_i, Z_j, C, tile_flag, the float64 column sums): on a small matrix (configs[1]) the fill
 * kernels were a seventh of the sweep.  Each entry: a 4-byte aligned pointer and a byte count that is a multiple of 4
 * (0 = unused).  clr may be NULL. */
#define ORIANA_CLEAR_MAX 8
typedef struct oriana_clear_list {
    void   *ptr[ORIANA_CLEAR_MAX];
    int64_t bytes[ORIANA_CLEAR_MAX];
} oriana_clear_list;
int oriana_factor_prep_pair_clear(float *FU, float *FV, const float *logU, const float *logV, const float *maskV,
                                  const int32_t *row_index_u, const int32_t *row_index_v,
                                  int64_t n, int64_t m, int64_t K, float *scratch, const oriana_clear_list *clr,
                                  void *stream);
/* [r5] The cell side prepared by the Gamma update that produced E[log U] (oriana_gamma_update_prep /
 * oriana_gamma_update_finalize_prep below: FU holds exp(E[log U] - row maximum) of every row, mu_u [n] the row maxima (NaN: the
 * row holds a NaN), upart [4 * nupart] the per-group partial statistics of the maxima): this call combines the statistics,
 * prepares the gene side from logV, overwrites the cell rows the centred validity test rejects, and zero-fills `clr` -- the
 * 2 x 4 K n bytes of reading E[log U] twice (statistics, preparation) and the second write of FU disappear from the sweep
 * (0.9 ms of 2.9 ms outside the passes at 1M x 30k, K = 100).  Same F and same statistics as oriana_factor_prep_pair up to the
 * summation order of the partials. */
int oriana_factor_prep_pair_fused(float *FU, const float *mu_u, const float *upart, int64_t nupart, float *FV,
                                  const float *logV, const float *maskV, const int32_t *row_index_v,
                                  int64_t n, int64_t m, int64_t K, float *scratch, const oriana_clear_list *clr, void *stream);
int64_t oriana_prep_scratch_bytes(void);   /* includes 4096 bytes for the log-sum centres at oriana_prep_center_offset() */
int64_t oriana_prep_center_offset(void);

/* ---- the responsibility pass ----------------------------------------------------------------
 * Replaces the loop nests  GaP.compute_Z_q_expectations        (oriana/models/gap.py:67-80)
 *                          ZIGaP.compute_Z_q_expectations      (oriana/models/zigap.py:79-95)
 *                          SparseGaP.compute_Z_q_expectations  (oriana/models/sparse_gap.py:81-97)
 *                          SparseZIGaP.compute_Z_q_expectations(oriana/models/sparse_zigap.py:100-116)
 * in three kernels over the tiled non-zeros:
 *   oriana_row_pass : s_ij = x_ij / sum_k FU[i,k] FVden[j,k]   and   R[i,:] = sum_j w_ij s_ij FVacc[j,:]
 *   oriana_col_pass : C[j,:] (+)= sum_i s_ij G[i,:]
 *   oriana_fixup    : entries the shifted form cannot represent (NaN sentinel in s) are evaluated
 *                     exactly as the reference does (expf of the sum, den > 0 guard) and added
 *                     with atomics to the dense outputs.
 */
int oriana_row_pass(const oriana_counts *cm,
                    const float *FU,        /* (n, Kp) */
                    const float *FV,        /* (m, Kp) */
                    const float *w_nz,      /* [rslots] per-entry weight (D_hat at the non-zeros) or NULL = 1 */
                    float *R,               /* (n, Kp) out -- left untouched when s_rs is given (the pass then only produces s) */
                    float *s_cs,            /* [cslots] out: s_ij (unweighted), column-side slots; padding slots must hold 0 */
                    float *sw_cs,           /* [cslots] out: w_ij s_ij (required iff w_nz) */
                    float *s_rs,            /* [rslots] out: s_ij, row-side slots, or NULL */
                    int32_t *tile_flag,     /* [nrb*ncb] out: 1 if the tile holds slow-path entries (zero it first) */
                    int64_t K, void *stream);

/* The plain row pass (w_nz = NULL, no row-side copy of s) with the gene tiles of every row block split over
 * `gene_splits` work-groups: a row block is one work-group, so a matrix of 10,000 cells (configs[1]) ran the pass on 40
 * of the 256 CUs.  R is then (gene_splits, n, Kp): every group stores the row sums of its gene range in its own slab
 * (no atomics, nothing to clear); oriana_finalize_slabs / oriana_gamma_update_finalize add the slabs up.
 * oriana_row_pass_gene_splits: the split that fills the chip for a SHORT matrix (below 256 row-side work-groups; 1 from
 * there on).  [r4] Long matrices: see oriana_row_pass_plan below, which splits the row blocks of the last round only. */
int64_t oriana_row_pass_gene_splits(const oriana_counts *cm, int64_t K);
int oriana_row_pass_split(const oriana_counts *cm, const float *FU, const float *FV, float *R, float *s_cs,
                          int32_t *tile_flag, int64_t K, int64_t gene_splits, void *stream);

/* The row pass of the sparse models with the S_hat-weighted sums folded in (sparse_gap.py:88-95): the dot product
 * runs against FV (= exp-shifted E[log V] masked by S_tilde), the accumulation against FV2 (= FV * S_hat), both staged
 * side by side in LDS -- R[i,:] = sum_j w_ij s_ij FV2[j,:] comes out of this pass and oriana_row_spmm is not needed.
 * Returns ORIANA_EKRANGE when the two images of 256 factor rows do not fit (Kp > 64): use oriana_row_pass (with s_rs)
 * followed by oriana_row_spmm then. */
int oriana_row_pass_masked(const oriana_counts *cm, const float *FU, const float *FV, const float *FV2,
                           const float *w_nz, float *R, float *s_cs, float *sw_cs, int32_t *tile_flag, int64_t K,
                           void *stream);

/* [r4] Which work-groups of the row pass share a row block.  The two-lane kernels (33 <= Kp <= 64, 85 <= K <= 100) run one
 * 512-thread group per CU, so the pass advances in ROUNDS of 256 row blocks and the last, partly filled round costs a whole
 * one (measured at 1M x 30k, K = 100: 3840 / 3907 / 4096 row blocks = 33.9 / 35.8 / 36.2 ms).  The row blocks [0, nfull) --
 * the full rounds -- are one work-group each (all gene tiles, row sums in slab 0 of R); every row block from nfull on is
 * `parts` work-groups: part p takes the gene tiles [edge[p], edge[p+1]) and stores its row sums in slab p, so the last round
 * is made of shorter groups.  nfull = 0 is the split of oriana_row_pass_split (short matrices); nfull = nrb, parts = 1 no
 * split at all.  R = (parts, n, Kp); the rows below 256 * nfull have slab 0 only (oriana_finalize_slabs_from). */
typedef struct oriana_row_split {
    int32_t nfull;
    int32_t parts;          /* 1 .. 8 */
    int32_t edge[9];        /* edge[0] = 0 <= ... <= edge[parts] = ncb */
} oriana_row_split;

/* The split that fills the chip for this matrix and K (what oriana_row_pass_gene_splits decided alone in round 3).
 * tile_cost: HOST array of ncb relative costs of the gene tiles (the ranges are cut at equal-cost points; genes are packed
 * by decreasing density, so equal tile counts are not equal work) or NULL = equal.  ORIANA_ROW_SPLIT_ROUNDS=off: only the
 * short-matrix rule. */
int oriana_row_pass_plan(const oriana_counts *cm, int64_t K, const double *tile_cost, oriana_row_split *out);

/* [r5] The same for a given number of compute units (oriana_row_pass_plan uses oriana_device_cus()): host code only, no
 * device call -- the plan of a 128-, 256- or 304-CU part can be formed (and tested) anywhere. */
int oriana_row_pass_plan_cus(const oriana_counts *cm, int64_t K, const double *tile_cost, int64_t cus, oriana_row_split *out);
/* Compute units of the device the calling thread has selected (hipDeviceAttributeMultiprocessorCount, cached per device;
 * the environment's ORIANA_CUS overrides; 256 where no device is visible): the "rounds of the chip" every plan counts in. */
int64_t oriana_device_cus(void);

/* [r4] Every variant of the row pass behind one entry: oriana_row_pass (FV2 = NULL) or oriana_row_pass_masked (FV2 given,
 * s_rs must be NULL; ORIANA_EKRANGE where the two images do not fit) under a split (NULL = none); s_rs given: R untouched.
 * Kernels other than the two-lane ones take nfull = 0 only (ORIANA_EINVAL otherwise) and cut their ranges evenly. */
int oriana_row_pass_general(const oriana_counts *cm, const float *FU, const float *FV, const float *FV2,
                            const float *w_nz, float *R, float *s_cs, float *sw_cs, float *s_rs, int32_t *tile_flag,
                            int64_t K, const oriana_row_split *split,
                            const float *den_min,   /* device: the den threshold (scratch of oriana_factor_prep_pair +
                                                       oriana_prep_den_threshold_offset()); NULL = the constant 1e-10 */
                            void *stream);
/* [r4] The den threshold of the shifted form: s = x / den' is trusted when den' >= threshold, below it the entry takes the
 * exact slow path (oriana_fixup).  The guarantee behind it is that the reference's own float32 den = exp(mu_i + mv_j) den'
 * is a normal number with room to spare (>= 3e-30); with the smallest sum mu_i + mv_j of the factor matrices at hand
 * (row statistics of oriana_factor_prep_pair) the threshold is 3e-30 exp(-sum_lo) in [1e-25, 1e-10] instead of the
 * worst-case constant 1e-10 -- ZI-pCMF drifts along U c, V / c and after 25 sweeps at configs[2] a third of the tiles
 * held entries between the two.  ORIANA_DEN_THRESHOLD=fixed keeps the constant. */
int64_t oriana_prep_den_threshold_offset(void);

/* R[i,:] = sum_j w_ij s_ij FV[j,:] with s given in row-side slots (sparse models: S_hat-weighted sums). */
int oriana_row_spmm(const oriana_counts *cm, const float *s_rs, const float *w_nz,
                    const float *FV, float *R, int64_t K, void *stream);

int oriana_col_pass(const oriana_counts *cm, const float *s_cs,
                    const float *G,         /* (n, Kp) */
                    float *C,               /* (m, Kp) accumulated with atomics: zero it first */
                    int64_t K,
                    /* optional work list [nwork][3] = (column block, first row block, end row block):
                     * one workgroup per item; a column block is oriana_col_block_tiles(K) adjacent column
                     * tiles; items should carry similar slot counts (genes differ widely in density).
                     * NULL / 0: uniform row bands. */
                    const int32_t *work, int64_t nwork,
                    void *stream);

/* Two products over ONE walk of the column-side stream: C1 += s G1 and C2 += s G2 (the sparse models' per-gene sums
 * and log sums, sparse_gap.py:96-97), both images of a row block side by side in LDS.  `work`: items
 * (column TILE, first row block, end row block) -- one tile per item, whatever oriana_col_block_tiles(K) says.
 * Returns ORIANA_EKRANGE when two images do not fit (Kp > 64): call oriana_col_pass twice then. */
int oriana_col_pass_dual(const oriana_counts *cm, const float *s_cs, const float *G1, const float *G2, float *C1,
                         float *C2, int64_t K, const int32_t *work, int64_t nwork, void *stream);
/* [r6] Analysis entry, not on the path of a sweep: the same sums with float64 accumulators in a fixed order and one rounding to
 * float32 at the end (C += f32(sum_i s_ij G_i)): the most a compensated accumulation of the float32 kernels could reach
 * (tools/parity_report.py: ORIANA_COL_F64=1 routes the per-gene sums of the sparse models' log sums through it). */
int oriana_col_pass_f64acc(const oriana_counts *cm, const float *s_cs, const float *G, float *C, int64_t K, void *stream);
/* Deterministic debug mode of the column pass (no counterpart in the reference, which is
 * single-threaded): every work item stores its accumulators in its own slab of `scratch`
 * (oriana_col_pass_det_scratch_bytes(K, nwork) bytes) instead of adding them to C with float atomics, and a
 * second kernel adds the slabs of each column block to C in work-item order.  Two runs give bit-identical C;
 * the default path differs from it only by the order of the float32 additions.  Needs a work list. */
int64_t oriana_col_pass_det_scratch_bytes(int64_t K, int64_t nwork);
int oriana_col_pass_det(const oriana_counts *cm, const float *s_cs, const float *G, float *C, int64_t K,
                        const int32_t *work, int64_t nwork, float *scratch, void *stream);

/* ---- the densest genes on the matrix cores (hybrid layout) --------------------------------------------------
 * Same loop nest (oriana/models/gap.py:67-80), other evaluation: the first `gd` PACKED genes (the densest ones, a
 * multiple of 32; the host picks them from a density threshold) are kept as a dense uint16 block instead of the
 * sliced non-zero layout; the oriana_counts of a hybrid layout then covers the packed genes [gd, m) only (its
 * col_perm, FV and C pointers are the full arrays advanced by gd rows).  Their share of the pass,
 *     den = FU FV^T,  s = x / den,  R += S FV,  C += S^T FU,
 * runs as matrix products on the bf16 matrix cores in float32-EQUIVALENT arithmetic: every float32 operand enters as
 * its exact three-way bf16 split (24 bits), six cross products per term, float32 accumulation, no sum longer than one
 * tile on the matrix core (csrc/dense_pass.hip; DESIGN.md section 7 has the measured error).  Entries whose
 * denominator fails the test of the sparse side get the same NaN sentinel and the same exact slow path.
 * Counts must be integers in [0, 65535) for a gene to be eligible (the host checks). */
typedef struct {
    int64_t n;             /* cells of this shard */
    int64_t gd;            /* dense genes = packed columns [0, gd), multiple of 32 */
    int64_t nct;           /* allocated cell tiles of 32 rows: 8 * ceil(n / 256) */
    const uint16_t *x;     /* [nct][gd / 32][1024] counts, register order of the row kernel (dense_pass.hip) */
} oriana_dense;

/* 1 if the dense evaluation is compiled for this K (K <= 100). */
int oriana_dense_supported(int64_t K);
/* 16-byte pieces of one 32-row tile's operand image; side 0 = gene side (FV), 1 = cell side (FU). */
int64_t oriana_dense_image_pieces(int64_t K, int side);
/* Pack `rows` rows (a multiple of 32 unless they are the last ones) of the already column-permuted dense chunk X
 * (columns [0, gd), leading dimension ldx, dtypes as oriana_pack_count) into cell tiles ct_first.. of xd. */
int oriana_dense_pack(const void *X, int xdtype, int64_t rows, int64_t gd, int64_t ldx, int64_t ct_first,
                      uint16_t *xd, void *stream);
/* Split operand images of a padded (rows, Kp) factor matrix, one per tile of 32 rows: img holds
 * ceil(rows / 32) * oriana_dense_image_pieces(K, side) * 16 bytes. */
int oriana_dense_images(void *img, const float *F, int64_t rows, int64_t K, int side, void *stream);
/* Row side: S (nct * gd/32 * 1024 floats, out) and R[i,:] += sum_j s_ij FV[j,:] over the dense genes (R must hold the
 * sparse genes' sums or zeros; added with atomics when gene_splits > 1).  flag: [nct][gd / 32] out (every entry written:
 * 1 where the tile of s holds sentinels). */
int oriana_dense_row_pass(const oriana_dense *d, const float *FU, const void *imgV, float *R, float *S,
                          int32_t *flag, int64_t K, int64_t gene_splits, void *stream);
/* [r4] ... with the 256-cell blocks from tail_nfull on split into tail_parts even gene-tile ranges, part p ADDING into slab p
 * of R = (tail_parts, n, Kp) (struct oriana_row_split: the same rows as the sliced row pass of a hybrid layout splits; gene_splits
 * must be 1 then).  tail_parts <= 1: oriana_dense_row_pass. */
int oriana_dense_row_pass_tail(const oriana_dense *d, const float *FU, const void *imgV, float *R, float *S,
                          int32_t *flag, int64_t K, int64_t gene_splits, int64_t tail_nfull, int64_t tail_parts,
                               const float *den_min /* as oriana_row_pass_general */, void *stream);
/* Gene side: C[j,:] += sum_i s_ij FU[i,:] for the dense genes (atomics: zero C first), imgU = the cell-side images. */
int oriana_dense_col_pass(const oriana_dense *d, const void *imgU, const float *S, float *C, int64_t K,
                          int64_t cell_splits, void *stream);
/* Exact slow path for the flagged tiles (adds to the dense outputs Z_hat_i, Z_hat_j; clears the sentinels in S). */
int oriana_dense_fixup(const oriana_dense *d, const int32_t *flag, float *S, const float *logU, const float *logV,
                       const int32_t *row_perm, const int32_t *col_perm, float *Z_hat_i, float *Z_hat_j, int64_t K,
                       void *stream);

/* The same with the D_hat[i, k] weight of zigap.py:94 on the gene side: Z_hat_j[j, k] += dq[i, k] r_ijk (dq: (n, K) float32,
 * caller's cell order; NULL = oriana_dense_fixup). */
int oriana_dense_fixup_weighted(const oriana_dense *d, const int32_t *flag, float *S, const float *logU, const float *logV,
                                const int32_t *row_perm, const int32_t *col_perm, float *Z_hat_i, float *Z_hat_j,
                                const float *dq, int64_t K, void *stream);
/* The slow path of every nest: Z_log (zigap.py:95, sparse_gap.py:97; may be NULL), the D_hat[i, k] weight (dq, may be NULL),
 * the masks of the sparse models (S_tilde, S_hat: both or neither; (m, K) float32, caller's gene order).  zj_packed != 0:
 * Z_hat_j is indexed by the PACKED gene index (the row-sharded pCMF sweep exchanges the per-gene sums in packed order). */
int oriana_dense_fixup_variant(const oriana_dense *d, const int32_t *flag, float *S, const float *logU, const float *logV,
                               const int32_t *row_perm, const int32_t *col_perm, float *Z_hat_i, float *Z_hat_j,
                               float *Z_log, const float *dq, const float *S_tilde, const float *S_hat, int64_t K,
                               int zj_packed, void *stream);
/* oriana_dense_images with the second image (the operand of the accumulation R += S F2 and its tail pieces) taken from F2:
 * the sparse models accumulate against FV * S_hat while den runs against the masked FV (sparse_gap.py:88-95). */
int oriana_dense_images2(void *img, const float *F, const float *F2, int64_t rows, int64_t K, int side, void *stream);
/* D_hat[i, j] = value at every non-zero count of the dense genes (zigap.py:77, 135); D_hat: (n, ld) float32, caller's
 * cell / gene order. */
int oriana_dense_fix_nz(const oriana_dense *d, float *D_hat, int64_t ld, const int32_t *row_perm, const int32_t *col_perm,
                        double value, void *stream);

/* Count statistics / deviance sums of the dense genes (the share of oriana_count_stats and oriana_metric_nnz that the
 * sliced layout of a hybrid matrix does not cover): colsum, colnnz (caller's gene order), out2 += {sum (x log x - x),
 * sum x^2}; with U, V (dense float64 (n, K), (m, K)): out4 += {sum Lambda, sum x log Lambda, sum Lambda^2,
 * sum x Lambda} over the non-zero counts, Lambda = U V^T in float64.  Any output may be NULL. */
int oriana_dense_metric(const oriana_dense *d, const double *U, const double *V, const int32_t *row_perm,
                        const int32_t *col_perm, double *colsum, double *colnnz, double *out2, double *out4,
                        int64_t K, void *stream);

/* Z[o,k] = (accumulate ? Z[o,k] : 0) + F[i,k] * R[i,k] (* mul[o,k] if mul), o = row_index ? row_index[i] : i
 * -- dense (r, K) out from padded (r, Kp) in. */
int oriana_finalize(float *Z, const float *F, const float *R, const float *mul, const int32_t *row_index,
                    int64_t r, int64_t K, int accumulate, void *stream);

/* Z[o,k] += F[i,k] * (R[0][i,k] + ... + R[nslab-1][i,k]) -- R = (nslab, r, Kp) as oriana_row_pass_split leaves it. */
int oriana_finalize_slabs(float *Z, const float *F, const float *R, int64_t nslab, const int32_t *row_index,
                          int64_t r, int64_t K, void *stream);
/* [r4] the same where only the rows from slab_row0 on have more than slab 0 (oriana_row_split: slab_row0 = 256 * nfull). */
int oriana_finalize_slabs_from(float *Z, const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                          int64_t r, int64_t K, void *stream);

/* Slow path (exact reference arithmetic) for the entries flagged by oriana_row_pass.
 * variant bit 0: S_tilde / S_hat present (sparse models); bit 1: D_hat weights (w_nz);
 * bit 2: reference quirk zigap.py:94 (dq = D_hat[:, :K] dense (n, K)).
 * Adds into Zi (n, K), Zj (m, K), Zlog (m, K) (any may be NULL) and rewrites the sentinels in
 * s_cs / sw_cs / s_rs as 0. */
int oriana_fixup(const oriana_counts *cm, const int32_t *tile_flag,
                 float *s_cs, float *sw_cs, float *s_rs,
                 const float *logU, const float *logV,
                 const float *S_tilde, const float *S_hat,
                 const float *w_nz, const float *dq,
                 float *Zi, float *Zj, float *Zlog,
                 int64_t K, int variant, void *stream);

/* ---- deviance / Frobenius metrics (base.py:58-87 with loglikelihood_X, sparse_zigap.py:44-51) --------
 * The metrics split into sums over the stored (non-zero) entries, sums over the zero entries and closed
 * forms of column sums (oriana_amd/models/base.py:_metric_terms spells the algebra out).
 *   oriana_factor_cast_f32 : F (r, Kp) f32 = float32(E[row_index[i], :] (* mul)) -- with these factors
 *                            oriana_row_pass leaves s_ij = x_ij / Lambda_ij in s_rs
 *   oriana_metric_nnz      : out4 += { sum Lambda, sum x log Lambda, sum Lambda^2, sum x Lambda } over the stored
 *                            entries (Lambda = x / s; entries with the NaN sentinel are recomputed in f64)
 *   oriana_count_stats     : constants of X: per-gene sums and non-zero counts, sum (x log x - x), sum x^2
 *   oriana_dropout_metric  : out2 += { sum log(pi_j exp(-Lambda_ij) + 1 - pi_j), sum Lambda_ij^2 } over the
 *                            entries with X == 0 and round(D_hat) == 1 (the mask of base.py:60-61, 67);
 *                            Lambda = U V^T on the f64 matrix cores, never stored
 */
int oriana_factor_cast_f32(float *F, const double *E, const float *mul, const int32_t *row_index,
                           int64_t r, int64_t K, void *stream);
int oriana_metric_nnz(const oriana_counts *cm, const float *s_rs, const double *U, const double *V,
                      int64_t K, double *out4, void *stream);
int oriana_count_stats(const oriana_counts *cm, double *colsum, double *colnnz, double *out2, void *stream);
int oriana_dropout_metric(double *out2, const float *D_hat, const double *U, const double *V, const double *pi_d,
                          const uint32_t *nzmask, int64_t n, int64_t m, int64_t K, void *stream);

/* ---- stateless drop-ins with the reference's exact signatures (outputs first) ------------------
 * One entry per loop nest, arguments in the reference's order, all matrices dense C-contiguous f32
 * on the device: X, D_hat (n, m); log_U_hat and the row-side output (n, K); log_V_hat, S_tilde, S_hat
 * and the gene-side outputs (m, K).  The callee zero-fills the outputs (gap.py:69-70).  `ws` is
 * caller-provided scratch of at least oriana_zq_workspace_bytes() bytes (256-byte aligned); these
 * entry points pack X on every call, as the reference re-casts X on every call (gap.py:94), and
 * synchronise the stream once (the slot totals size the record arrays).  The model classes use the
 * resident oriana_counts instead.
 *   oriana_zq_gap_f32          GaP.compute_Z_q_expectations          oriana/models/gap.py:67-80
 *   oriana_zq_zigap_f32        ZIGaP.compute_Z_q_expectations        oriana/models/zigap.py:79-95
 *                              reference_quirks != 0 keeps the D_hat[i, k] index of zigap.py:94
 *                              (needs K <= m); 0 weights the per-gene sums with D_hat[i, j]
 *   oriana_zq_sparse_gap_f32   SparseGaP.compute_Z_q_expectations    oriana/models/sparse_gap.py:81-97
 *   oriana_zq_sparse_zigap_f32 SparseZIGaP.compute_Z_q_expectations  oriana/models/sparse_zigap.py:100-116
 */
int64_t oriana_zq_workspace_bytes(int64_t n, int64_t m, int64_t K, int64_t nnz_bound);
int oriana_zq_gap_f32(float *Z_hat_i, float *Z_hat_j,
                      const float *log_U_hat, const float *log_V_hat, const float *X,
                      int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes, void *stream);
int oriana_zq_zigap_f32(float *DZ_hat_i, float *DZ_hat_j, float *DZ_exp_logsum_hat,
                        const float *log_U_hat, const float *log_V_hat, const float *D_hat, const float *X,
                        int64_t n, int64_t m, int64_t K, int reference_quirks,
                        void *ws, int64_t ws_bytes, void *stream);
int oriana_zq_sparse_gap_f32(float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                             const float *log_U_hat, const float *log_V_hat,
                             const float *S_tilde, const float *S_hat, const float *X,
                             int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes, void *stream);
int oriana_zq_sparse_zigap_f32(float *DSZ_hat, float *DZ_hat, float *DZ_exp_logsum_hat,
                               const float *log_U_hat, const float *log_V_hat,
                               const float *S_tilde, const float *S_hat, const float *D_hat, const float *X,
                               int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes, void *stream);

/* ---- Gamma / Bernoulli updates and the M-step --------------------------------------------------
 * oriana_gamma_update: one side (U or V) of update_variational_parameters
 *   (gap.py:96-110, zigap.py:114-128, sparse_gap.py:117-132, sparse_zigap.py:137-152) fused with
 *   Gamma.mean / Gamma.meanlog (nodes/probabilistic/gamma.py:37-61):
 *     a1 = max(1e-15, nan_to_num(prior1[k] + zmul[i,k] * Z[i,k]))
 *     a2 = max(1e-15, nan_to_num(prior2[k] + rmul[i,k] * (rate_mat ? rate_mat[i,k] : rate_vec[k])))
 *     E = a1 / a2 (f64);  Elog = f32(digamma(f32(a1))) - logf(f32(a2))
 *   and the column sums sum_i E[i,k], sum_i Elog[i,k] (f64, ADDED into colsum_E / colsum_Elog:
 *   zero them first; they feed the M-step means, gap.py:120-122, and the other side's rate).
 *   Z may be NULL (initialisation: a1/a2 are taken as they are, only expectations are produced).
 */
int oriana_gamma_update(double *a1, double *a2, double *E, float *Elog,
                        double *colsum_E, double *colsum_Elog,
                        const double *prior1, const double *prior2,
                        const float *Z, const float *zmul,
                        const double *rate_vec, const double *rate_mat, const float *rmul,
                        int64_t r, int64_t K, void *stream);

/* The same update with the last step of the responsibility pass folded in (pCMF, gap.py:79-80 + 96-110): rows are walked
 * in PACKED order p, Z[o,k] += F[p,k] * sum_s R[s][p,k] with o = row_index ? row_index[p] : p (what oriana_finalize_slabs
 * does; R = (nslab, r, Kp); Z stays a complete output) and a1 = prior1 + Z, a2 = prior2 + rate_vec at once -- one launch
 * and one pass over Z less per side. */
int oriana_gamma_update_finalize(double *a1, double *a2, double *E, float *Elog,
                                 double *colsum_E, double *colsum_Elog,
                                 const double *prior1, const double *prior2,
                                 float *Z, const float *F, const float *R, int64_t nslab, const int32_t *row_index,
                                 const double *rate_vec, int64_t r, int64_t K, void *stream);
/* [r4] ... with oriana_finalize_slabs_from's slab_row0. */
int oriana_gamma_update_finalize_from(double *a1, double *a2, double *E, float *Elog,
                                 double *colsum_E, double *colsum_Elog,
                                 const double *prior1, const double *prior2,
                                 float *Z, const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                 const double *rate_vec, int64_t r, int64_t K, void *stream);

/* [r5] Both updates with the cell-side half of the NEXT sweep's factor preparation folded in (FU_next != NULL; see
 * oriana_factor_prep_pair_fused): FU_next (r, Kp) float32, zero-filled once by the caller (padding columns are never written),
 * gets exp(Elog - row maximum) in the packed row order of F / R (finalize form) or in the caller's row order (plain form: only
 * for counts without a row permutation); mu_out [r]; upart [4 * oriana_gamma_update_prep_blocks(r, K)].  The blocks query
 * returns 0 when no vector kernel serves this K (odd K above 64, K above 256): the caller then keeps the separate preparation.
 * FU_next = NULL: exactly oriana_gamma_update / oriana_gamma_update_finalize_from. */
int64_t oriana_gamma_update_prep_blocks(int64_t r, int64_t K);
int oriana_gamma_update_prep(double *a1, double *a2, double *E, float *Elog,
                             double *colsum_E, double *colsum_Elog,
                             const double *prior1, const double *prior2,
                             const float *Z, const float *zmul,
                             const double *rate_vec, const double *rate_mat, const float *rmul,
                             int64_t r, int64_t K, float *FU_next, float *mu_out, float *upart, void *stream);
int oriana_gamma_update_finalize_prep(double *a1, double *a2, double *E, float *Elog,
                                 double *colsum_E, double *colsum_Elog,
                                 const double *prior1, const double *prior2,
                                 float *Z, const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                 const double *rate_vec, int64_t r, int64_t K, float *FU_next, float *mu_out, float *upart,
                                 void *stream);
/* [r6] The finalize form for pCMF's cell side WITHOUT the two (r, K) float64 matrices no kernel of a sweep reads:
 *   gap.py:98   a2[i, :] = max(1e-15, alpha2 + sum_j V_hat)   -- the same K numbers in every row: written ONCE, to a2_row[K];
 *   gap.py:101  U_hat = a1 / a2 (Gamma._mean, gamma.py:37-46)  -- not stored; its column sums go to colsum_E as before.
 * The launch moves 3.2 instead of 4.8 GB at configs[3].  The caller materialises a2 / U_hat on access (oriana_amd/models/gap.py:
 * a float64 broadcast and a float64 division, bit-identical to what the kernel would have stored).  Everything else as
 * oriana_gamma_update_finalize_prep.  ORIANA_EKRANGE when no vector kernel serves this K / these pointers (odd K above 64,
 * K above 256, a pointer that is not 16-byte aligned): take oriana_gamma_update_finalize_prep. */
int oriana_gamma_update_finalize_lazy(double *a1, double *a2_row, float *Elog,
                                 double *colsum_E, double *colsum_Elog,
                                 const double *prior1, const double *prior2,
                                 float *Z, const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                 const double *rate_vec, int64_t r, int64_t K, float *FU_next, float *mu_out, float *upart,
                                 void *stream);

/* M-step for one Gamma node (gap.py:117-129; utils.py:39-51):
 *   p1 = max(1e-15, nan_to_num(inverse_digamma(log(p2) + f32(colsum_Elog / count))))
 *   p2 = max(1e-15, nan_to_num(p1 / (colsum_E / count)))            (K-vectors, f64, in place) */
int oriana_mstep_gamma(double *p1, double *p2, const double *colsum_E, const double *colsum_Elog,
                       double count, int64_t K, void *stream);

/* Both Gamma nodes in one launch.  keep_v (2K values or NULL): copy of the V side's column sums {sum E, sum Elog} --
 * a sweep that accumulates them in scratch (cleared with the other scratch of the sweep) keeps them here for the next
 * sweep's cell-side rate, sum_j V_hat (gap.py:98). */
int oriana_mstep_gamma_pair(double *p1u, double *p2u, const double *colsum_E_u, const double *colsum_Elog_u, double count_u,
                            double *p1v, double *p2v, const double *colsum_E_v, const double *colsum_Elog_v, double count_v,
                            double *keep_v, int64_t K, void *stream);

/* Column sums of a dense (r, K) f64 matrix, optionally times an f32 (r, K) multiplier, added
 * into out[K] (zero it first) -- `V_hat.sum(axis=0)`, gap.py:98. */
int oriana_colsum_f64(double *out, const double *A, const float *mul, int64_t r, int64_t K, void *stream);

/* ---- zero-inflated / sparse model pieces -------------------------------------------------------
 * D_q update (zigap.py:130-136, sparse_zigap.py:163-169), in three steps:
 *   oriana_dropout_update : p_d = sigmoid(logit(pi_d)[None, :] - Lambda), columns with pi_d <= 0 -> 1e-10,
 *                           pi_d >= 1 -> 1 - 1e-10; D_hat = float32(p_d).  Lambda = U_hat V_hat^T, dense
 *                           (rows, m) f64, may alias p_d.
 *   oriana_dropout_fix_nz : p_d[X != 0] = value (1 - 1e-10; 1.0 at initialisation, zigap.py:77) and D_hat,
 *                           from the tiled layout.
 *   oriana_colsum_wide_f64: out[j] += sum_i A[i, j]  (pi_d = mean(p_d, axis=0), zigap.py:158).
 */
int oriana_dropout_update(double *p_d, float *D_hat, const double *Lambda, const double *pi_d,
                          /* optional: bit mask of X != 0 (oriana_nzmask_f32 layout) -> the override
                           * p_d[X != 0] = 1 - 1e-10 is applied in the same pass */
                          const uint32_t *nzmask,
                          /* optional: colsum[j] += sum_i p_d[i, j] (zero it first) */
                          double *colsum,
                          int64_t rows, int64_t m, void *stream);
/* mask[(i / 32) * m + j] bit (i % 32) = (D[i, j] != 0); ceil(rows / 32) * m words. */
int oriana_nzmask_f32(uint32_t *mask, const float *D, int64_t rows, int64_t m, void *stream);
/* The same update with Lambda = U_hat V_hat^T formed on the matrix cores inside the kernel (f64 MFMA),
 * so that Lambda never goes through HBM: U (n, K), V (m, K) f64 row-major, K <= 256.  p_d and D_hat
 * are optional outputs (NULL: not stored) -- the models keep only D_hat resident and evaluate p_d on
 * access from a snapshot of (U, V, pi_d). */
int oriana_dropout_update_fused(double *p_d, float *D_hat, const double *U, const double *V, const double *pi_d,
                                const uint32_t *nzmask, double *colsum, int64_t n, int64_t m, int64_t K,
                                void *stream);
/* Rate terms of the ZI Gamma updates on the matrix cores, D_hat (n, m) f32 read in place and promoted
 * to f64 as np.dot does:  trans = 0: out[n, K] += D_hat W[m, K]   (zigap.py:116, np.dot(D_hat, V_hat))
 *                         trans = 1: out[m, K] += D_hat^T W[n, K] (zigap.py:124, np.dot(D_hat.T, U_hat))
 * `out` must be initialised (zeros for a plain product).  K <= 256. */
int oriana_dense_times_factor(double *out, const float *D, const double *W, int64_t n, int64_t m, int64_t K,
                              int trans, void *stream);
/* The dense work of one ZI SWEEP on the float32 matrix cores (csrc/dense_f32.hip), K <= 128 (else ORIANA_EKRANGE:
 * use the float64 entries above).  Long sums leave the matrix core every 256 terms and end in float64; measured
 * against the float64 entries: D_hat within 2 float32 ulp, rate terms within 3e-7 relative.
 *
 * oriana_dropout_sweep_fused: D_hat[n, m] = f32(sigmoid(logit(pi_d) - U V^T)) with the overrides of zigap.py:130-136
 * (nzmask optional as above), colsum[j] += sum_i p_d[i, j] (optional, zero it first), and -- V_next / DV_next both
 * given or both NULL -- DV_next[n, K] += D_hat V_next[m, K] from the tile of D_hat still in registers: with V_next the
 * factor the NEXT sweep's cell update multiplies (zigap.py:116: the V_hat just updated; sparse_zigap.py:138:
 * S_hat * Vprime_hat) that sweep has no pass over D_hat left on the cell side.  DV_next must be zeroed first.
 * scratch: oriana_dropout_sweep_scratch_floats(m, K) floats of device scratch (logit(pi_d), float32 copies of V and
 * V_next).  arithmetic: how the float32 products are evaluated --
 *   ORIANA_MATRIX_F32     v_mfma_f32_32x32x2_f32 (a chain of single-rounding float32 FMAs);
 *   ORIANA_MATRIX_BF16X3  each float32 operand split into three bf16, six cross products on the bf16 matrix cores,
 *                         float32 accumulation (K <= 100 when the gene count is a multiple of 4, K <= 64 otherwise;
 *                         larger K silently take the float32 instruction; csrc/dense_f32.hip, csrc/dense_zi.hip): the same
 *                         error as the float32 chain on sums of <= 512 terms (which is all the kernel forms before it
 *                         leaves the matrix core), 2.7 x its rate. */
#define ORIANA_MATRIX_F32     0
#define ORIANA_MATRIX_BF16X3  1
int oriana_dropout_sweep_fused(float *D_hat, const double *U, const double *V, const double *pi_d,
                               const uint32_t *nzmask, double *colsum, const double *V_next, double *DV_next,
                               float *scratch, int arithmetic, int64_t n, int64_t m, int64_t K, void *stream);
int64_t oriana_dropout_sweep_scratch_floats(int64_t m, int64_t K);
/* [r6] The same sweep with the non-zero mask ALSO as per-lane flags for the K = 33 .. 100 bf16 x 3 kernel (csrc/dense_zi.hip,
 * dn::k_zi_row): every lane of a wave reads the 16 flags of the 16 values it holds with one 2-byte LDS read per tile and
 * applies the override p_d[X != 0] = 1 - 1e-10 (zigap.py:135) with two instructions per entry (bit-field extract, bit-field
 * insert) where the gene-major words of oriana_nzmask_f32 took three plus eight LDS reads per tile.
 * nztiles: oriana_nzmask_tiles_words(n, m) uint32 words (16-byte aligned), built ONCE per count matrix by oriana_nzmask_tiles
 * from the oriana_nzmask_f32 layout: per (cell tile of 32, gene tile of 32) 64 x 16 bits, bit v of entry l =
 * (X[32 ct + l % 32, 32 gt + 8 (v / 4) + 4 (l / 32) + v % 4] != 0); cell tiles padded to a multiple of 8.
 * nztiles == NULL (= oriana_dropout_sweep_fused): that kernel does not apply; K <= 64 takes the bf16 kernels of
 * csrc/dense_f32.hip, larger K the float32 matrix instruction. */
int oriana_dropout_sweep_fused_tiles(float *D_hat, const double *U, const double *V, const double *pi_d,
                                     const uint32_t *nzmask, const uint32_t *nztiles, double *colsum, const double *V_next,
                                     double *DV_next, float *scratch, int arithmetic, int64_t n, int64_t m, int64_t K,
                                     void *stream);
int64_t oriana_nzmask_tiles_words(int64_t n, int64_t m);
int oriana_nzmask_tiles(uint32_t *tiles, const uint32_t *nzmask, int64_t n, int64_t m, void *stream);
/* out[m, K] += D_hat^T W[n, K] (zigap.py:124), D_hat streamed once; `out` must be initialised.  arithmetic as above;
 * scratch: oriana_dense_t_scratch_floats(n, K) floats (16-byte aligned; the bf16 operand images of W), may be NULL
 * with ORIANA_MATRIX_F32. */
int oriana_dense_t_times_factor_f32(double *out, const float *D, const double *W, float *scratch, int arithmetic,
                                    int64_t n, int64_t m, int64_t K, void *stream);
int64_t oriana_dense_t_scratch_floats(int64_t n, int64_t K);
/* either output may be NULL */
int oriana_dropout_fix_nz(const oriana_counts *cm, double *p_d, float *D_hat, double value, void *stream);
/* the same with an explicit leading dimension of p_d / D_hat (the sliced part of a hybrid layout: cm->m counts its own genes only) */
int oriana_dropout_fix_nz_ld(const oriana_counts *cm, double *p_d, float *D_hat, double value, int64_t ld, void *stream);
int oriana_colsum_wide_f64(double *out, const double *A, int64_t rows, int64_t m, void *stream);
/* the same for a float32 matrix (D_hat, while p_d == D_hat exactly: zigap.py:77) */
int oriana_colsum_wide_f32(double *out, const float *A, int64_t rows, int64_t m, void *stream);
/* dq[i, k] = D[i, k], k < K: the columns the reference's zigap.py:94 reads (D_hat[i, k]). */
int oriana_take_cols_f32(float *out, const float *D, int64_t rows, int64_t m, int64_t K, void *stream);

/* S_q update (sparse_gap.py:134-141, sparse_zigap.py:154-161):
 *   p_s = nan_to_num(sigmoid(logit(pi_s)[:, None] - (-Zlog + nan_to_num(c * Vprime_hat)))),
 *   c = c_vec[k] (sum_i U_hat) or c_mat[j, k] (D_hat^T U_hat); rows with pi_s <= 0 / >= 1 overridden;
 *   S_hat = float32(p_s).  oriana_threshold_f32: S_tilde = (p_s > tau) (sparse_gap.py:113).
 *   oriana_rowmean_f64: pi_s = mean(p_s, axis=1) (sparse_gap.py:165). */
int oriana_sparsity_update(double *p_s, float *S_hat, const double *pi_s, const float *Zlog,
                           const double *c_vec, const double *c_mat, const double *Vprime_hat,
                           int64_t m, int64_t K, void *stream);
int oriana_threshold_f32(float *out, const double *p, double tau, int64_t len, void *stream);
int oriana_rowmean_f64(double *out, const double *A, int64_t r, int64_t K, void *stream);

/* out = A * B element-wise, f64 x f32 -> f64 (V_hat = S_hat * Vprime_hat, sparse_gap.py:118). */
int oriana_mul_f64_f32(double *out, const double *A, const float *B, int64_t len, void *stream);

/* Fout[i, :] = Fin[i, :] * mul[row_index ? row_index[i] : i, :]  (padded factor times a dense (., K) f32
 * matrix: S_hat-, D_hat[:, :K]- or E[log U]-weighted factors for the extra sums of the ZI / sparse loop
 * nests).  zero_guard: entries with Fin == 0 stay 0 whatever mul holds. */
int oriana_scale_factor(float *Fout, const float *Fin, const float *mul, const int32_t *row_index,
                        int64_t r, int64_t K, int zero_guard, void *stream);
/* The log sums sum_i r_ijk (lu_ik + lv_jk) (zigap.py:95, sparse_gap.py:97) as two column sums around a per-factor
 * centre a_k = mean of E[log U]_ik over the contributing cells (without it the two sums are each |lu| times larger than
 * their total once the sweeps have drifted, and the sparsity posterior amplifies the lost float32 digits):
 *   oriana_log_center            acc[0..K) = sum w logF, acc[K..2K) = sum w over the entries that carry weight (F > 1e-20,
 *                                finite log); w = W[., k] (dense (r, K) f32 in the caller's row order: the cell's Z_hat_i) or 1;
 *                                acc is zeroed here
 *   oriana_scale_factor_centered Fout = Fin * (mul - a_k), 0 where Fin == 0             (acc NULL: a = 0)
 *   oriana_finalize_zlog         Zlog[o, k] += FV[j, k] * (C2[j, k] + (logV[o, k] + a_k) * C[j, k]), o = row_index ? row_index[j] : j,
 *                                combined in float64 */
int oriana_log_center(double *acc, const float *F, const float *logF, const float *W, const int32_t *row_index,
                      int64_t r, int64_t K, void *stream);
int oriana_scale_factor_centered(float *Fout, const float *Fin, const float *mul, const double *acc,
                                 const int32_t *row_index, int64_t r, int64_t K, void *stream);
int oriana_finalize_zlog(float *Zlog, const float *FV, const float *C2, const float *C, const float *logV,
                         const double *acc, const int32_t *row_index, int64_t r, int64_t K, void *stream);

/* Element-wise special functions on f64 vectors (oriana/utils.py:9-15, 31-51) -- used by tests
 * and by the host mirror of oriana.utils. */
int oriana_digamma_f64(double *y, const double *x, int64_t len, void *stream);
int oriana_trigamma_f64(double *y, const double *x, int64_t len, void *stream);
int oriana_inverse_digamma_f64(double *y, const double *x, int64_t len, void *stream);
int oriana_sigmoid_f64(double *y, const double *x, int64_t len, void *stream);
int oriana_logit_f64(double *y, const double *x, int64_t len, void *stream);

/* ---- [r5] planning of the resident layout as plain C (host arrays in and out, no device) -------------------------------
 * What the kernels of a sweep depend on beyond the packed records.  engine.py (the Python host) and the resident handle below
 * both call these; tests/test_plan.py checks them on the CPU against a NumPy restatement.
 *
 * oriana_plan_gene_order: the internal gene order.  Genes in decreasing order of their non-zero count (col_nnz, summed over
 * all row shards; ties in the caller's order).  dense_density > 0 (hybrid layout): the genes expressed in at least that share
 * of the n_total cells whose counts all fit a uint16 block (bad[j] == 0: no negative, >= 65535 or non-integer entry; bad may be
 * NULL) come first, cut to a multiple of 32 (*gd), unless they hold less than min_share of the non-zeros (then *gd = 0). */
int oriana_plan_gene_order(const int64_t *col_nnz, const int64_t *bad, int64_t m, int64_t n_total,
                           double dense_density, double min_share, int32_t *order /* out [m] */, int64_t *gd /* out */);
/* oriana_plan_col_work: the work list of oriana_col_pass -- items (column block, first row block, end row block) of about
 * equal cost, ordered by row band (concurrent items stage the same factor rows).  tile_iters [nrb * ncb]: longest column-side
 * slice of each tile in iterations of 64 slots; width = oriana_col_block_tiles(K); cost of a row block of a column block =
 * 1.45 * (longest slice over its tiles; sum_price != 0: their sum) + 3.2 (staging 256 factor rows; microseconds on MI355X).
 * target_items = 0: 25-50 tiles per item, 9 to 36 items per CU; rounds != 0: re-cut so that the count lands just below a
 * multiple of `cus` (one 1024-thread group per CU: a partly filled last round costs a whole one).  items: out [cap][3],
 * cap >= oriana_plan_col_work_capacity(nrb, ncb, width). */
int64_t oriana_plan_col_work_capacity(int64_t nrb, int64_t ncb, int64_t width);
int oriana_plan_col_work(const int32_t *tile_iters, int64_t nrb, int64_t ncb, int64_t width, int64_t cus,
                         int64_t target_items, int rounds, int sum_price, int32_t *items, int64_t cap, int64_t *n_items);
/* Splits of the dense-gene kernels of a hybrid layout: gene ranges of oriana_dense_row_pass (at most two work-groups per CU)
 * and cell ranges of oriana_dense_col_pass (four per CU, a multiple of 8 cell tiles). */
int oriana_plan_dense_splits(int64_t n, int64_t gd, int64_t cus, int64_t *gene_splits, int64_t *cell_splits);

/* ---- [r5] resident handle: the count matrix packed ONCE for a host that is not Python --------------------------------------
 * The reference calls its loop nest once per step() with the same X (oriana/models/gap.py:89-94, zigap.py:105-112,
 * sparse_gap.py:107-115, sparse_zigap.py:126-135) and re-casts X every time; the stateless entries above repack X per call.
 * A handle owns (hipMalloc) the packed layout -- gene order, sliced records, the dense uint16 block of a hybrid layout
 * (dense_density > 0 and oriana_dense_supported(K); <= 0: sliced only), the column work list, the row split, the scratch of a
 * call -- built by the planning functions above for oriana_device_cus() compute units.  X: dense (n, ldx) float32 on the
 * DEVICE, or CSR on the HOST (indptr [n + 1], indices, data; expanded on the device in row chunks of at most 512 MB, so
 * neither side ever holds the dense matrix; duplicate entries add up).  The create calls synchronise the stream; the zq
 * calls are asynchronous on it, take DEVICE matrices in the reference's argument order (outputs first, zero-filled by the
 * callee, 2-D C-contiguous float32) and must not run concurrently on one handle. */
typedef struct oriana_resident oriana_resident;
int oriana_counts_create_dense_f32(oriana_resident **out, const float *X, int64_t n, int64_t m, int64_t ldx, int64_t K,
                                   double dense_density, void *stream);
int oriana_counts_create_csr(oriana_resident **out, const int64_t *indptr, const int32_t *indices, const float *data,
                             int64_t n, int64_t m, int64_t K, double dense_density, void *stream);
int oriana_counts_destroy(oriana_resident *h);
/* info[0..12] = {n, m, K, Kp, non-zeros, dense genes, row-side slots, column-side slots, resident bytes, column work items,
 * row-split parts, first split row block, compute units planned for} */
int oriana_counts_info(const oriana_resident *h, int64_t *info, int64_t len);
/* GaP.compute_Z_q_expectations (gap.py:67-80) on the resident layout, sliced or hybrid. */
int oriana_zq_gap_resident(oriana_resident *h, float *Z_hat_i, float *Z_hat_j, const float *log_U_hat, const float *log_V_hat,
                           void *stream);
/* The three twins (zigap.py:79-95, sparse_gap.py:81-97, sparse_zigap.py:100-116) on the resident layout.
 *
 * [r6] oriana_counts_declare_unit_dropout(h, 1): the caller declares that every D_hat it will pass is exactly 1 wherever the
 * count is non-zero -- which holds for every D_hat the reference's own models produce (zigap.py:135 / sparse_zigap.py:168 set
 * p_d[X != 0] = 1 - 1e-10, Bernoulli.mean casts to float32: bernoulli.py:45; the initial p_d = (X > 0): zigap.py:77).  The
 * nests then run exactly as oriana_amd's model classes run them: sliced OR hybrid handle, no gather of D_hat (it is read for
 * the D_hat[i, k] weights of zigap.py:94 only), and for Kp <= 64 the fused kernels -- the S_hat-weighted row sums out of the
 * two-image row pass, both per-gene sums out of one dual column pass (Kp > 64: den-only row pass + second row product + two
 * column passes).  oriana_zq_sparse_gap_resident has no D_hat and always runs this way.
 * Without the declaration (the default) D_hat may hold anything: it is gathered at the stored entries on every call and the
 * weighted four-kernel form runs, on a SLICED handle only (ORIANA_EUNIT for a hybrid one: the dense-gene kernels carry no
 * per-entry weights).
 * The third output (the log sums) may be NULL: the reference's ZIGaP never reads it (zigap.py:105-112) and the model classes
 * skip it; the sparse models need it (sparse_gap.py:135). */
int oriana_counts_declare_unit_dropout(oriana_resident *h, int on);
int oriana_zq_zigap_resident(oriana_resident *h, float *DZ_hat_i, float *DZ_hat_j, float *DZ_exp_logsum_hat,
                             const float *log_U_hat, const float *log_V_hat, const float *D_hat, int reference_quirks, void *stream);
int oriana_zq_sparse_gap_resident(oriana_resident *h, float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                                  const float *log_U_hat, const float *log_V_hat, const float *S_tilde, const float *S_hat, void *stream);
int oriana_zq_sparse_zigap_resident(oriana_resident *h, float *DSZ_hat, float *DZ_hat, float *DZ_exp_logsum_hat,
                                    const float *log_U_hat, const float *log_V_hat, const float *S_tilde, const float *S_hat,
                                    const float *D_hat, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ORIANA_HIP_H */
