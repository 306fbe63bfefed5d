/*
 * text_alignment_amd.h -- C ABI of libta_hip.so, the MI355X (gfx950) hot path of
 * DDMAL/text_alignment.
 *
 * The reference has no FFI on this path: its boundary is two Python calls and one shell
 * command.  Each entry point below names the reference interface
 * whose arithmetic it replaces; the Python mirror of that interface lives in
 * text_alignment_amd/ and binds these symbols with ctypes.
 *
 * Conventions: plain pointers and sizes only; every pointer marked [dev] is a device
 * (HBM) address owned by the caller; [host] is host memory.  `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Calls enqueue work and return without
 * synchronising.  Return value: 0 = ok, negative = TA_E*; ta_last_error() describes the
 * most recent failure on the calling thread.  The library keeps no mutable global state besides the
 * lazily loaded code object and one-time, thread-safe raises of kernels' dynamic-LDS limits.
 */
#ifndef TEXT_ALIGNMENT_AMD_H
#define TEXT_ALIGNMENT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TA_OK 0
#define TA_EINVAL (-1)    /* bad argument (null pointer, negative size, unsupported shape) */
#define TA_ERANGE (-2)    /* scores would overflow the integer fast path */
#define TA_EHIP (-3)      /* a HIP runtime call failed */
#define TA_ELIMIT (-4)    /* problem exceeds a kernel limit (LDS capacity) */

/* bit flags for ta_nw_batch */
#define TA_NW_FILL 1u
#define TA_NW_TRACEBACK 2u
#define TA_NW_CODES8 4u      /* caller asserts every token id < 255: 1-byte codes in LDS (ta_nw2_batch) */
#define TA_NW_WIDE 8u        /* ta_nw_batch: force one problem over several workgroups (HBM hand-off rows) */
#define TA_NW_NARROW 16u     /* ta_nw_batch: force one workgroup per problem; default: wide iff nprob < 256 */
#define TA_NW_OPENS_SAME 32u /* ta_nw2_batch hint: gap_open_x == gap_open_y in every scoring system of the batch */
/* ta_nw2_batch, score profile in LDS.  TA_NW_ALPHABET(a) with 1 <= a <= 254 makes two statements about the
 * batch, of different kinds:
 *   - a caller ASSERTION, like TA_NW_CODES8: every token id of every problem is < a.  It is not checked.  The
 *     OCR ids index the per-wave profile directly (row pitch 256 B, a + 1 rows), so an id >= a reads another
 *     wave's table or LDS beyond the profile and ids >= 256 wrap: the alignment comes out silently WRONG,
 *     not slower.  Derive `a` from the data (max id + 1), as text_alignment_amd.textSeqCompare.NWBatch does,
 *     or add TA_NW_CHECK_IDS while bringing a caller up.
 *   - HINTS that change speed only: every gap open is <= 0 and match/mismatch minus both gap extends fit a
 *     signed byte.  A problem that does not meet them (and, under TA_NW_OPENS_SAME, one whose two gap opens
 *     differ) is still aligned correctly, through the kernel's general cell.
 * Leave the field 0 (no profile: token ids are compared per cell) when the id bound is not known. */
#define TA_NW_ALPHABET_SHIFT 8
#define TA_NW_ALPHABET(a) (((uint32_t)(a) & 0xFFu) << TA_NW_ALPHABET_SHIFT)
/* ta_nw2_batch launch-shape overrides (tests, A/B timing): phase 1 without the score profile even where
 * the hints allow it; phase 1 with exactly w waves per workgroup (1, 2, 4 or 8; ignored when the tallest
 * problem has fewer strips or the LDS does not hold it; 0 = the library's own choice). */
/* ta_nw_batch launch-shape override (tests, timing): rows per lane of the one-pass fill, 1, 2 or 4 (0 = the
 * library's choice: 2 for batches too small to give every SIMD a 256-row strip, else 4; 1 -- 64-row
 * strips -- measured slower than 2 at every size tried and is never chosen).  Fill and traceback of
 * one batch must be given the same value (and the same nprob / max_n), also when issued as two calls. */
#define TA_NW_ROWS_SHIFT 20
#define TA_NW_ROWS(r) (((uint32_t)(r) & 0x7u) << TA_NW_ROWS_SHIFT)
/* ta_nw2_batch, traceback launch shape (tests, timing): waves per problem of phase 2 -- 1 (one wave walks the chunks
 * one after the other), 2 or 4 (the chunks along the path dealt to the waves of a workgroup, each re-filling the
 * chunk it expects ahead of the walk), 3 (TWO problems per wave, 32 lanes each, walking back half-strips of 128 rows:
 * large batches), 5 / 6 (both: two / four waves per pair of problems on half-strips: medium batches); 0 = the library's
 * choice by batch size.  Same results either way. */
#define TA_NW_TBWAVES_SHIFT 24
#define TA_NW_TBWAVES(w) (((uint32_t)(w) & 0x7u) << TA_NW_TBWAVES_SHIFT)
#define TA_NW_NO_PROFILE 64u
/* ta_nw2_batch, debug guard for the two caller assertions above (TA_NW_CODES8: ids < 255; TA_NW_ALPHABET(a): ids < a):
 * with this bit the call first checks every token id of the batch on the device, WAITS for the result (one synchronisation
 * of `stream` -- its scratch word is a stream-ordered allocation, so other streams of the device run on: not for timed code) and returns TA_EINVAL -- nothing else launched -- if an id breaks an assertion
 * the flags make.  Without it a violated assertion gives silently wrong alignments. */
#define TA_NW_CHECK_IDS 128u
#define TA_NW_WAVES_SHIFT 16
#define TA_NW_WAVES(w) (((uint32_t)(w) & 0xFu) << TA_NW_WAVES_SHIFT)

int ta_version(void);
const char* ta_last_error(void);
/* PCI address ("0000:c1:00.0", NUL-terminated, len >= 13) of HIP device `device` -- the key under /sys/bus/pci/devices/
 * from which a rank of the page-sharded job (the loop of alignToOCR.py:407-438, one process per GPU) reads the NUMA node
 * of its GPU to bind itself next to it.  [host] */
int ta_device_pci_bus_id(int32_t device, char* out, int32_t len);
/* n host arrays into one staging buffer in one call: nbytes[k] bytes from src[k] to dst + dst_off[k] (memcpy; all pointers
 * [host]).  How a chunk's text-line strips reach the page-locked buffer they cross PCIe from -- where the reference writes
 * each strip to a PNG file for the recogniser (alignToOCR.py:131-132) -- without the interpreter lock being taken per strip. */
int ta_host_copy_pieces(void* dst, const void* const* src, const int64_t* dst_off, const int64_t* nbytes, int32_t n);
/* Characters and boxes of every decoded line of a batch -- the loop of alignToOCR.py:160-182 over all lines at once, host
 * arithmetic: entry i of line b is (dec_t, dec_c)[dec_off[b] + i], i < dec_n[b]; its position x = (t - pad) * raw_w[b] /
 * (T[b] - 2 pad) goes through the `.llocs` text's one decimal ("%.1f") and int(np.round(x + x_min[b])) (half to even); a box
 * runs from the previous character's position (x_min[b] for the first) to its own, between y_min[b] and y_max[b]; classes
 * with cps[c] < 0 ('~' and '', alignToOCR.py:175) are dropped but still move the edge.  dec_len = entries in dec_t / dec_c (a
 * line whose dec_off + dec_n exceeds it is refused: TA_EINVAL).  Outputs have capacity sum(dec_n):
 * out_line, out_cp, out_boxes [k][4] = ulx, uly, lrx, lry; *out_count = characters kept.  All pointers [host]. */
/* Union of the OCR character boxes under every syllable of a batch of pages (alignToOCR.py:285-324 after the alignment;
 * host arithmetic).  ops: uint8 alignment columns of all pages end to end (0 pair, 1 transcript character over a gap, 2 gap
 * over an OCR character); idx[nidx]: row of boxes ([nboxes][4] = ulx, uly, lrx, lry) of the character of every OCR-carrying
 * column, in order; syllable s = transcript characters first_t[s] .. last_t[s] of the concatenated transcripts (ranges
 * disjoint, ascending).  out_low[s] = largest uly under the syllable (INT64_MIN: no OCR character under it -- the reference
 * skips it, :313-314); out_box[s][4] = union of the boxes whose uly is that value (the lower of two text lines, :318-320).
 * All pointers [host]. */
int ta_host_syllable_boxes(const uint8_t* ops, int64_t ncol, const int64_t* idx, int64_t nidx, const int64_t* boxes,
                           int64_t nboxes, const int64_t* first_t, const int64_t* last_t, int64_t nsyl,
                           int64_t* out_low, int64_t* out_box);
int ta_host_chars_of_batch(const int32_t* dec_t, const int32_t* dec_c, const int64_t* dec_n, const int64_t* dec_off,
                           const int64_t* T, const int64_t* raw_w, const int64_t* x_min, const int64_t* y_min,
                           const int64_t* y_max, const int64_t* cps, int32_t ncps, int32_t pad, int32_t nlines,
                           int64_t dec_len, int64_t* out_line, int64_t* out_cp, int64_t* out_boxes, int64_t* out_count);

/*
 * Affine-gap Needleman-Wunsch, replaces textSeqCompare.perform_alignment
 * (reference textSeqCompare.py:13-177; called from alignToOCR.py:273).
 *
 * ta_nw_workspace_bytes: bytes of pointer-matrix workspace one n x m problem needs
 * (1 byte per DP cell plus the skew padding of the strip layout -- 63 columns per strip, sized for the
 * finest strips TA_NW_ROWS can ask for -- plus one 8(m+2)-byte hand-off row per four strips for the wide
 * launch; multiple of 1024).
 */
int64_t ta_nw_workspace_bytes(int32_t n, int32_t m);

/* largest m (OCR-side length) the LDS-resident hand-off row supports */
int32_t ta_nw_max_m(void);

/*
 * ta_nw_batch: fill (textSeqCompare.py:53-88) and/or traceback (textSeqCompare.py:96-170)
 * of `nprob` independent problems.
 *
 *   t_codes, o_codes [dev]  concatenated int32 token ids of all transcripts / OCR strings;
 *                           equal ids <=> equal tokens (textSeqCompare.py:32); ids < 65536
 *   t_off, o_off     [dev]  int64[nprob+1] prefix offsets into the code arrays
 *   params           [dev]  int32 scoring systems, 6 per system: match, mismatch,
 *                           gap_open_x, gap_open_y, gap_extend_x, gap_extend_y
 *                           (textSeqCompare.py:30-34); params_stride = 0 -> one shared
 *                           system, 6 -> one per problem (the parameter grid of
 *                           evaluate_text_alignment.py:181-198)
 *   ws               [dev]  pointer-matrix workspace; ws_off [dev] int64[nprob] byte offset of
 *                           each problem's region (16-byte aligned, ta_nw_workspace_bytes long)
 *   ops_out          [dev]  alignment columns; problem p owns ops_off[p] .. ops_off[p] + n_p + m_p
 *                           and its alignment is RIGHT-aligned in that region: the last
 *                           ops_len[p] bytes, forward order, 0 = (t,o) pair, 1 = (t,'_'),
 *                           2 = ('_',o)  (textSeqCompare.py:115-164 after the reversal at :167)
 *   ops_off          [dev]  int64[nprob]
 *   ops_len          [dev]  int32[nprob] alignment lengths (written by the traceback; ta_nw2_batch sets them to -1
 *                           first, and a problem whose length is still negative afterwards was not walked)
 *   max_n, max_m            host-side maxima of the problem sizes (sizes LDS and the grid)
 *   score_bound             host-side bound on (max_n + max_m + 2) * max|param| used for the
 *                           overflow check (TA_ERANGE if it does not fit 2^23)
 *   flags                   TA_NW_FILL | TA_NW_TRACEBACK
 */
int ta_nw_batch(const int32_t* t_codes, const int64_t* t_off,
                const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                const int32_t* params, int32_t params_stride,
                uint8_t* ws, const int64_t* ws_off,
                uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len,
                int32_t max_n, int32_t max_m, int64_t score_bound,
                uint32_t flags, void* stream);

/*
 * ta_nw2_batch: the same aligner in two phases -- a score-only wavefront fill that leaves
 * checkpoints, then a traceback that re-derives pointers only in windows around the path
 * (csrc/ta_nw2.hip).  Arguments and results exactly as ta_nw_batch, except that `ws` regions are
 * ta_nw2_workspace_bytes(n, m) long (lane-state checkpoints every 16 groups + the bottom rows of every
 * half-strip of 128 rows: ~0.3 B per cell instead of the 1 B/cell pointer matrix).  TA_NW_FILL = phase 1,
 * TA_NW_TRACEBACK = phase 2.
 */
int64_t ta_nw2_workspace_bytes(int32_t n, int32_t m);
/* widest OCR string ta_nw2_batch takes (its LDS holds the OCR codes only; the hand-off rows between
 * strips live in the workspace) */
int32_t ta_nw2_max_m(void);
int ta_nw2_batch(const int32_t* t_codes, const int64_t* t_off,
                 const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                 const int32_t* params, int32_t params_stride,
                 uint8_t* ws, const int64_t* ws_off,
                 uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len,
                 int32_t max_n, int32_t max_m, int64_t score_bound,
                 uint32_t flags, void* stream);

/* What phase 2 (the traceback) of ta_nw2_batch launches for a batch of nprob problems: 1, 2 or 4 = waves per problem
 * (nw_trace2_kernel / nw_trace2w_kernel), 3 = two problems per wave on half-strips (nw_trace2h_kernel: batches that
 * fill the chip and share one scoring system), 5 / 6 = two / four waves per pair of problems on half-strips
 * (nw_trace2hw_kernel: medium batches under one scoring system).  Pure host function. */
int32_t ta_nw2_traceback_plan(int32_t nprob, int32_t params_stride, uint32_t flags);
/* What phase 1 of ta_nw2_batch would launch for a batch whose tallest / widest problem is
 * max_n x max_m under `flags` (the hints above): out[0] = 1 compare-select cell, 2 score profile
 * in LDS; out[1] = waves per workgroup; out[2] = dynamic LDS bytes per workgroup; out[3] = 1 if the
 * single-gap-open form of the cell is used.  For a batch large enough to fill the chip (the launch
 * also weighs the number of problems when it picks out[1]).  Pure host function (no GPU call). */
int ta_nw2_phase1_plan(int32_t max_n, int32_t max_m, uint32_t flags, int32_t* out);
/* the same for a batch of nprob problems: exactly what ta_nw2_batch will launch */
int ta_nw2_phase1_plan_batch(int32_t max_n, int32_t max_m, int32_t nprob, uint32_t flags, int32_t* out);

/*
 * ta_nw_general: the same aligner for scoring systems the integer kernel does not take --
 * a caller-supplied scoring function (textSeqCompare.py:27-29; the host tabulates it over
 * the distinct tokens into `table`, row-major [t id][o id], row length tm) or non-integral
 * numbers.  IEEE float64 throughout, bit-identical to the reference.  One problem per call.
 *
 *   t, o      [dev]  int32 token ids, n and m of them
 *   params    [dev]  6 doubles: match, mismatch, gap_open_x, gap_open_y, gap_extend_x, gap_extend_y
 *   table     [dev]  optional (NULL = use match/mismatch)
 *   score_ws  [dev]  ta_nw_general_score_bytes(n) bytes
 *   ptr_ws    [dev]  ta_nw_general_ptr_bytes(n, m) bytes
 *   ops_out   [dev]  n + m bytes, alignment right-aligned as in ta_nw_batch; ops_len [dev] int32
 */
int64_t ta_nw_general_score_bytes(int32_t n);
int64_t ta_nw_general_ptr_bytes(int32_t n, int32_t m);
int ta_nw_general(const int32_t* t, int32_t n, const int32_t* o, int32_t m,
                  const double* params, const double* table, int32_t tm,
                  double* score_ws, uint8_t* ptr_ws,
                  uint8_t* ops_out, int32_t* ops_len, void* stream);
/* The same for many problems in one launch (one workgroup each), match/mismatch scoring only:
 * concatenated codes with offsets as in ta_nw_batch; params = double[nprob or 1][6] with
 * params_stride 6 or 0; score_off (in doubles) / ptr_off / ops_off (in bytes) = per-problem offsets
 * into the workspaces (sizes as above) and into ops_out (capacity n + m, right-aligned). */
int ta_nw_general_batch(const int32_t* t_codes, const int64_t* t_off,
                        const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                        const double* params, int32_t params_stride,
                        double* score_ws, const int64_t* score_off,
                        uint8_t* ptr_ws, const int64_t* ptr_off,
                        uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len, void* stream);

/*
 * Line recogniser: replaces the `ocropus-rpred` subprocess of
 * alignToOCR.perform_ocr_with_ocropus (reference alignToOCR.py:142-147; arithmetic of the
 * third-party ocropy 1.3.3).  All pointers [dev].
 *
 * Lines are concatenated row-wise: line b owns rows row_off[b] .. row_off[b] + T[b] of
 *   x     [rows][48]   prepared line (ink = 1, 16 zero columns of padding each side)
 *   hout  [rows][200]  BiLSTM outputs [forward 100 | reversed LSTM flipped back 100]
 *   probs [rows][no]   softmax outputs;  logits [rows][no] optional (NULL to skip)
 *
 * ta_lstm_forward: group_lines = int32[ngroups][16] line ids (-1 = empty slot; [ngroups][4] in mode 2); one
 *   workgroup runs the lines of a group in lockstep, so groups should hold lines of similar length.
 *   mode 0: exact f32 MFMA chain.  wp = ta_lstm_packed_weight_floats(0) floats: B fragments
 *          [dir 2][wave 7][gate GI,GF,GO,CI][k-step 38][lane 64] =
 *          W_gate[unit 16*wave + lane%16][kp 4*kstep + lane/16], kp: 0 bias, 1..48 x,
 *          49..51 zero, 52..151 h; units >= 100 zero.
 *   mode 1: 16-bit matrix cores on split operands, f32 accumulation: W = W_hi (bf16) + W_r (fp16 of
 *          W - W_hi), activations three bf16 terms + one fp16; W.a ~ W_hi.(a_lo + a_mid + a_hi) + W_r.a16.
 *          wp = ta_lstm_packed_weight_floats(1) 4-byte units holding 16-bit patterns
 *          [dir 2][wave 7][plane hi,r][gate 4][k-step 5][lane 64][8] =
 *          plane of W_gate[unit 16*wave + lane%16][kp 32*kstep + 8*(lane/16) + j], kp as above
 *          padded with zeros to 160.
 *   mode 2: mode 0's arithmetic, bit for bit, on groups of FOUR lines (group_lines = int32[ngroups][4]):
 *          v_mfma_f32_4x4x1_16B_f32, one k per instruction -- a step costs a quarter of mode 0's, for the
 *          batches (a page, a few hundred lines) whose time is their longest line's.  wp =
 *          ta_lstm_packed_weight_floats(2) floats [dir 2][wave 7][k 152][lane 64] =
 *          W_gate(lane % 4)[unit 16*wave + lane/4][kp k], kp and padding as in mode 0.
 *   peep = float[2][3][112]: WIP, WFP, WOP per direction, units >= 100 zero.
 *   h0, c0, tstart (all NULL for fresh lines, or all given): a "line" may be the continuation of a
 *          sequence run elsewhere -- float h0[lines][2][100] / c0[lines][2][100] are the LSTM output
 *          and cell state before its first step, per direction (index 1 = the reversed LSTM, whose
 *          first step is the line's LAST row), and int32 tstart[lines][2] the number of steps of the
 *          sequence already done (> 0: the peepholes that ocropy skips at t = 0 apply from the start).
 * ta_lstm_output: w2p = float[201][16*ceil(no/16)] (classes beyond `no` zero): row 0 = bias
 *   column W2[:, 0]; row 1 + 4*kk + kq = W2[:, 1 + 50*kq + kk] (kk < 50, kq < 4) -- the k order
 *   in which the kernel consumes a row of hout.  no <= 128.  probs / logits / summary are each
 *   optional (at least one of probs, summary): summary = float[rows][4] =
 *   {P(class 0), best P, best class (integer bits), 0}, all the decoder needs.
 * ta_lstm_output_split: the same layer on the 16-bit matrix cores with split operands (the output
 *   stage of mode 1 above; products exact to ~2^-19 relative): w2s = ta_lstm_output_split_weight_bytes(no)
 *   bytes of 16-bit patterns [plane hi,r][class tile ceil(no/16)][k-step 7][lane 64][8] =
 *   plane of W2[16*tile + lane%16][1 + 32*kstep + 4*(lane/16) + 16*(j/4) + j%4] (inputs >= 200 and classes >= no
 *   zero), W2 = bf16 hi + fp16 rest as for the recurrence; bias = float[16*ceil(no/16)] = W2[:, 0].
 * ta_decode / ta_decode_summary: translate_back(outputs, threshold) per line, from the full
 *   probabilities or from the summaries; line b writes dec_n[b] (t, class) pairs at
 *   dec_t/dec_c + dec_off[b] (capacity (T[b] + 1) / 2 entries).
 */
int32_t ta_lstm_packed_weight_floats(int32_t mode);
int ta_lstm_forward(const float* x, const int64_t* row_off, const int32_t* T,
                    const int32_t* group_lines, int32_t ngroups,
                    const float* wp, const float* peep, float* hout, int32_t mode,
                    const float* h0, const float* c0, const int32_t* tstart, void* stream);
/*
 * The recurrence in FLOAT64 (csrc/ta_lstm_f64.hip) -- the arithmetic type of the reference's recogniser (ocropy
 * computes in float64 numpy; call site alignToOCR.py:142-147).  Two calls per run of groups:
 *
 * ta_lstm_xproj_f64: the input projection of every row at once, gx[dir][r][16 (unit / 4) + 8 (gate / 2) + 2 (unit % 4) + gate % 2] =
 *   W_gate[unit][0 .. 48] . [1; x[r]] in float64 for the `rows` rows at x; gx = ta_lstm_f64_gx_bytes(rows) bytes
 *   ([2][rows][400] doubles; the buffer is private to the pair of calls, its layout is the recurrence kernel's).  wx = ta_lstm_f64_weight_doubles(1) doubles [dir 2][tile 25][slot 13][lane 64]:
 *   slots 0..11 = W_gate(2 (i / 8) + i % 2)[unit 4 tile + (i % 8) / 2][1 + 4 slot + lane / 16], i = lane % 16 (the weights of
 *   x[4 slot + lane / 16]: B fragments of v_mfma_f64_16x16x4_f64, column i of a tile = its position in a row of gx);
 *   slot 12 = the column's bias W_gate(..)[unit ..][0] in every lane of the column (what the accumulators start from).
 * ta_lstm_forward_f64: the recurrence over `ngroups` groups of 16 lines (group_lines = int32[ngroups][16], -1 =
 *   empty slot) whose rows lie in [gx_row0, gx_row0 + gx_rows) -- row_off / hout use ABSOLUTE rows, gx holds the
 *   projection of rows gx_row0 .. only.  wh = ta_lstm_f64_weight_doubles(0) doubles
 *   [dir 2][wave 4][slot 7][k-step 25][lane 64] = W_gate(i / 4)[unit 4 tile + i % 4][49 + 4 kstep + lane / 16],
 *   tile = 6 wave + slot for slots 0..5 and 24 for slot 6 (the tile the waves split along k); peep = double[2][3][100]: WIP, WFP, WOP per direction.
 *   hout [rows][200] float (the float64 outputs rounded once).  h0 / c0 (double[lines][2][100]) / tstart as in
 *   ta_lstm_forward.  status (optional, [dev] one int32 the caller zeroes): the kernel ORs TA_LSTM_F64_PARTS_LATE into
 *   it if a workgroup's bounded wait for the partial sums of the tile its waves share ever runs out (the outputs of
 *   that group are NaN from there on); the launch itself still returns TA_OK -- read the word back with the results.
 * ta_lstm_forward_f64_g4: the same recurrence, bit for bit the same outputs, over groups of FOUR lines (group_lines =
 *   int32[ngroups][4]) on v_mfma_f64_4x4x4_4b_f64: a quarter of the cost per step, for batches whose 16-line groups
 *   would not fill the GPU or would wait for their longest line.  wh4 = ta_lstm_f64_weight_doubles(3) doubles
 *   [dir 2][tile 25][k-step 25][lane 64] = W_gate(lane % 4)[unit 4 tile + (lane / 4) % 4][49 + 4 kstep + lane / 16];
 *   every other argument as in ta_lstm_forward_f64.
 */
#define TA_LSTM_F64_PARTS_LATE 1
int64_t ta_lstm_f64_weight_doubles(int32_t which);
int64_t ta_lstm_f64_gx_bytes(int64_t rows);
int ta_lstm_xproj_f64(const float* x, int64_t rows, const double* wx, double* gx, void* stream);
int ta_lstm_forward_f64(const double* gx, int64_t gx_row0, int64_t gx_rows, const int64_t* row_off,
                        const int32_t* T, const int32_t* group_lines, int32_t ngroups, const double* wh,
                        const double* peep, float* hout, const double* h0, const double* c0,
                        const int32_t* tstart, int32_t* status, void* stream);
int ta_lstm_forward_f64_g4(const double* gx, int64_t gx_row0, int64_t gx_rows, const int64_t* row_off,
                           const int32_t* T, const int32_t* group_lines, int32_t ngroups, const double* wh4,
                           const double* peep, float* hout, const double* h0, const double* c0,
                           const int32_t* tstart, int32_t* status, void* stream);
int ta_lstm_output(const float* y, int64_t rows, const float* w2p, int32_t no,
                   float* probs, float* logits, float* summary, void* stream);
int64_t ta_lstm_output_split_weight_bytes(int32_t no);
int ta_lstm_output_split(const float* y, int64_t rows, const void* w2s, const float* bias, int32_t no,
                         float* probs, float* logits, float* summary, void* stream);
int ta_decode_summary(const float* summary, const int64_t* row_off, const int32_t* T,
                      int32_t nlines, float threshold,
                      int32_t* dec_t, int32_t* dec_c, int32_t* dec_n, const int64_t* dec_off,
                      void* stream);
int ta_decode(const float* probs, const int64_t* row_off, const int32_t* T,
              int32_t nlines, int32_t no, float threshold,
              int32_t* dec_t, int32_t* dec_c, int32_t* dec_n, const int64_t* dec_off,
              void* stream);

/*
 * ta_rows_gather: prepared rows that already lie in device memory into the recogniser's row layout -- line b's T[b]
 *   rows of 48 floats are copied from the device ADDRESS src[b] (16-byte aligned; any allocation, e.g. the device copy
 *   of a page-locked block of rows that ONE transfer brought over as it was) to rows dst_row[b] .. of x.  max_T >= every
 *   T[b].  All pointers [dev].  It stands where the reference writes each strip to a PNG file for the recogniser
 *   (alignToOCR.py:131-132): the hand-over of a batch's line images, without a host-side copy per line.
 */
int ta_rows_gather(const int64_t* src, const int64_t* dst_row, const int32_t* T, int32_t nlines,
                   int32_t max_T, float* x, void* stream);

/*
 * Line normaliser: what `ocropus-rpred` does to each PNG strip of alignToOCR.py:131-147 before the
 * network sees it -- ocropy 1.3.3 CenterNormalizer.measure / dewarp / normalize and prepare_line
 * (third-party arithmetic).  All pointers [dev].
 *
 * Strips are uint8 greyscale images (white background), concatenated: strip b = pix + pix_off[b],
 * hh[b] rows of ww[b] pixels.  gw holds the gaussian kernels; gw_off[b][3] are the offsets of the
 * CENTRE taps and gr[b][3] the radii of the three kernels of strip b (sigma 0.5 h along rows,
 * 1.0 h along columns, 0.3 h for the centre line; scipy's truncate = 4 kernels, computed by the
 * caller so that they are bit-identical to the host's).  ws: 3 * hh * ww doubles per strip at
 * ws_off[b]; arg / center: ww[b] ints per strip at col_off[b]; minmax: 2 ints per strip.
 *
 * ta_linenorm_measure writes center (the smoothed, truncated centre line), r_out (half-height of
 * the band cut around it) and wout (normalised width int(48 / (2 r) * w)) -- the caller reads wout
 * back to size the outputs.  ta_linenorm_resample writes the recogniser's input rows: strip b owns
 * rows row_off[b] .. row_off[b] + wout[b] + 32 of x [rows][48] (16 zero rows of padding each
 * side); tmp: 48 * wout[b] floats per strip at tmp_off[b]; omax: one word per strip.
 */
int ta_linenorm_measure(const uint8_t* pix, const int64_t* pix_off, const int32_t* hh,
                        const int32_t* ww, int32_t nlines, const double* gw,
                        const int64_t* gw_off, const int32_t* gr, double* ws,
                        const int64_t* ws_off, int32_t* arg, int32_t* center,
                        const int64_t* col_off, int32_t* minmax, int32_t* r_out,
                        int32_t* wout, void* stream);
int ta_linenorm_resample(const uint8_t* pix, const int64_t* pix_off, const int32_t* hh,
                         const int32_t* ww, int32_t nlines, const int32_t* center,
                         const int64_t* col_off, const int32_t* minmax, const int32_t* r,
                         const int32_t* wout, float* tmp, const int64_t* tmp_off,
                         uint32_t* omax, float* x, const int64_t* row_off, void* stream);

/*
 * Page preprocessing primitives: the full-page image passes of textAlignPreprocessing.py:160-285
 * (Gamera's to_onebit, despeckle, cc_analysis, rotation_angle_projections, rotate,
 * filter_short_runs / filter_narrow_runs, projection_rows in the reference).  One page at a time;
 * images are uint8 planes [h][w] (ink = 1).  All pointers [dev].
 *
 * ta_pp_label: 8-connected components; lab[p] = linear index of the component's first pixel in
 *   raster order, -1 on background; stats = int32[5][h*w] = area, x0, y0, x1, y1, indexed by that
 *   root (only the entries of roots are written; the others keep whatever the buffer held); flag = one device int (unused since round 3: tiles are stitched by one lock-free union-find
 *   pass, nothing iterates and nothing waits for the stream).
 * ta_pp_components: up to cap records {root, area, x0, y0, x1, y1} (any order), *count = true number.
 * ta_pp_filter_components: clears components with area < min_area or more than max_height rows.
 * ta_pp_angle_histograms: hist[a][row] of the page decimated by `step` and rotated by angle a
 *   (cos_sin = {cos a0, sin a0, ...}): pixel (y, x) lands on row rint(cy + dy cos a - dx sin a).
 * ta_pp_rotate: (bilinear resample of the 0/1 plane through output -> input map
 *   in = M out + offset, mo = {m00, m01, m10, m11, off0, off1}, zeros outside) > 0.5.
 * ta_pp_open_runs: morphological opening with a line of `len` pixels along `axis`.
 */
int ta_pp_histogram(const uint8_t* img, int64_t n, uint32_t* hist256, void* stream);
int ta_pp_threshold(const uint8_t* img, int64_t n, int32_t thr, int32_t invert, uint8_t* ink, void* stream);
int ta_pp_label(const uint8_t* ink, int32_t h, int32_t w, int32_t* lab, int32_t* stats, int32_t* flag,
                void* stream);
/* ta_pp_label for nimg images with shared waits: ink / lab / stats are [host] arrays of [dev] pointers,
 * h / w [host] arrays, flags nimg [dev] ints */
int ta_pp_label_batch(int32_t nimg, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                      int32_t* const* lab, int32_t* const* stats, int32_t* flags, void* stream);
int ta_pp_components(const int32_t* lab, const int32_t* stats, int32_t h, int32_t w, int32_t* recs,
                     int32_t cap, int32_t* count, void* stream);
int ta_pp_filter_components(uint8_t* ink, const int32_t* lab, const int32_t* stats, int32_t h, int32_t w,
                            int32_t min_area, int32_t max_height, void* stream);
int ta_pp_invert(uint8_t* ink, int64_t n, void* stream);
int ta_pp_angle_histograms(const uint8_t* ink, int32_t h, int32_t w, int32_t step, const double* cos_sin,
                           int32_t nang, uint32_t* hist, void* stream);
/* The skew search from a list of the page's ink pixels, made once for every angle of both sweeps:
 * ta_pp_ink_points: points[i] = (row << 16) | column on the grid decimated by `step` (room for
 * ceil(h/step) * ceil(w/step) entries; at most 65535 rows / columns), in no particular order; *count [dev].
 * ta_pp_angle_histograms_points(points, count, hs = ceil(h/step), ws = ceil(w/step), ...): the histograms of
 * ta_pp_angle_histograms, count for count. */
int ta_pp_ink_points(const uint8_t* ink, int32_t h, int32_t w, int32_t step, uint32_t* points, uint32_t* count,
                     void* stream);
int ta_pp_angle_histograms_points(const uint32_t* points, const uint32_t* count, int32_t hs, int32_t ws,
                                  const double* cos_sin, int32_t nang, uint32_t* hist, void* stream);
int ta_pp_rotate(const uint8_t* ink, int32_t h, int32_t w, uint8_t* out, int32_t oh, int32_t ow,
                 const double* mo, void* stream);
int ta_pp_open_runs(const uint8_t* in, uint8_t* out, int32_t h, int32_t w, int32_t len, int32_t axis,
                    void* stream);
int ta_pp_row_sums(const uint8_t* ink, int32_t h, int32_t w, int32_t* sums, void* stream);
int ta_pp_clear_rows(uint8_t* ink, int32_t w, const int32_t* rows, int32_t nrows, void* stream);
/* Host arithmetic, no device work: for the candidate rows idx[0..k) (local maxima) of a row projection
 * d[0..n), the arguments of the logarithms of calculate_peak_prominence (textAlignPreprocessing.py:59-110):
 * arg[c] = d[i] where d[i] == data_max, else d[i] - min(d[lo:hi]) + 1 over the reference's slice between i
 * and the nearest strictly higher sample.  All pointers [host]. */
int ta_pp_peak_prominence_args(const double* d, int32_t n, const int32_t* idx, int32_t k, double data_max,
                               double* arg);
/* The text-line strips of a page, cut from its ink plane (h x w, non-zero = ink) into one packed buffer as
 * the greyscale images the reference writes for the recogniser (alignToOCR.py:131-132; ink 0 on 255).
 * boxes (device): nstrips x {ulx, uly, lrx, lry, byte offset of the strip in out}, inclusive corners inside
 * the plane (the caller's to guarantee); strip s is row-major (lry - uly + 1) x (lrx - ulx + 1) at out + offset. */
int ta_pp_cut_strips(const uint8_t* ink, int32_t h, int32_t w, const int64_t* boxes, int32_t nstrips,
                     uint8_t* out, void* stream);

/*
 * Whole STAGES of the page preprocessing for a batch of n pages, one call each: everything between two of the
 * pipeline's data-dependent host decisions (Otsu threshold | skew sweeps | projection peaks | component selection |
 * strips) -- reference textAlignPreprocessing.py:160-285, where every page pass is a Gamera call.  A page's stage is a
 * dozen launches; made one by one from the host language they cost more host time than the kernels take.  Arguments
 * named like arrays are [host] arrays of n [dev] pointers / sizes; everything is enqueued on `stream`, nothing is
 * waited for; per page the kernels and their order are those of the single-page entry points above.
 *
 * The two stages that need connected components (ta_pp_binarise_batch, ta_pp_line_components_batch) find them over
 * RUNS since round 6 (a text page's ~6 M pixels are a few hundred thousand horizontal runs; a wave per row finds
 * them, one union-find pass joins each with the touching runs of the row above, 8-connectivity; a component's root is
 * its raster-first run): same components, areas, boxes and root pixels as ta_pp_label gives, a fifth of the memory
 * traffic; the holes are filled from the runs of PAPER, without inverting the plane and back.  lab[i] / stats[i] are
 * then scratch for the run tables (pages under 8 columns, or flags & TA_PP_LABEL_PIXELS: the per-pixel form).
 *
 * ta_pp_histogram_batch: hist [dev] = uint32[n][256] of img[i][0..npix[i]).
 * ta_pp_binarise_batch (:167-186): ink[i] = img[i] thresholded at thr[i], despeckled (components under `despeckle`
 *   pixels; ink, then background), components taller than max_height rows dropped; points[i] / counts[i] [dev uint32[n]]
 *   = ta_pp_ink_points of the page decimated by step[i].  lab[i] / stats[i]: h*w and 5*h*w int32 of scratch.
 * ta_pp_angle_histograms_points_batch: ta_pp_angle_histograms_points per page (cos_sin[i]: 2*nang[i] doubles [dev],
 *   hist[i]: uint32[nang[i]][hs[i]] [dev]).
 * ta_pp_deskew_batch (:187-195, :212-215): out[i] (oh[i] x ow[i]) = ink[i] rotated through mo[i] (ta_pp_rotate;
 *   mo[i] NULL: a copy, sizes equal); eroded[i] = out[i] opened with runs of runs_len pixels along the rows, then the
 *   columns, `rounds` times (runs_len <= 1: a copy; tmp[i]: oh*ow bytes of scratch); sums[i][r] = ink of eroded row r.
 * ta_pp_line_components_batch (:216-252): work[i] = eroded[i] with rows[i][0..nrows[i]) cleared, labelled (scratch as
 *   above); page i's component table (ta_pp_components) at recs [dev] + i*cap*6, its true count in counts[i] [dev].
 * ta_pp_cut_strips_batch: ta_pp_cut_strips per page into ONE packed buffer (the boxes carry their offsets).
 */
int ta_pp_histogram_batch(int32_t n, const uint8_t* const* img, const int64_t* npix, uint32_t* hist, void* stream);
#define TA_PP_LABEL_PIXELS 1   /* flags of the two stages that label: a label per PIXEL (ta_pp_label_batch) instead of
                                * connected components over RUNS -- the form of rounds 3-5, kept as the cross-check */
int ta_pp_binarise_batch(int32_t n, const uint8_t* const* img, const int32_t* h, const int32_t* w, const int32_t* thr,
                         int32_t despeckle, int32_t max_height, uint8_t* const* ink, int32_t* const* lab,
                         int32_t* const* stats, const int32_t* step, uint32_t* const* points, uint32_t* counts,
                         int32_t flags, void* stream);
int ta_pp_angle_histograms_points_batch(int32_t n, const uint32_t* const* points, const uint32_t* counts,
                                        const int32_t* hs, const int32_t* ws, const double* const* cos_sin,
                                        const int32_t* nang, uint32_t* const* hist, void* stream);
int ta_pp_deskew_batch(int32_t n, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                       const double* const* mo, uint8_t* const* out, const int32_t* oh, const int32_t* ow,
                       uint8_t* const* tmp, uint8_t* const* eroded, int32_t runs_len, int32_t rounds,
                       int32_t* const* sums, void* stream);
int ta_pp_line_components_batch(int32_t n, const uint8_t* const* eroded, const int32_t* h, const int32_t* w,
                                const int32_t* const* rows, const int32_t* nrows, uint8_t* const* work,
                                int32_t* const* lab, int32_t* const* stats, int32_t* recs, int32_t cap,
                                int32_t* counts, int32_t flags, void* stream);
int ta_pp_cut_strips_batch(int32_t n, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                           const int64_t* const* boxes, const int32_t* nstrips, uint8_t* packed, void* stream);

/*
 * Host arithmetic between the preprocessing stages ([host] pointers only, no device work): the per-page numpy passes
 * of the reference's Python between two Gamera calls, as plain loops that hold no interpreter lock; each reproduces
 * the numpy expression it replaces operation by operation (tests/test_preprocessing.py fuzzes them against numpy).
 * ta_host_otsu_batch: Otsu's threshold (first maximum of the between-class variance) of n 256-bin histograms.
 * ta_host_sharpest_rows: per page k the row of its nang[k] x hs[k] int32 histograms (at hist + off[k]) with the
 *   largest np.var (numpy's pairwise summation), and whether the page has a count at all; var_out may be NULL.
 * ta_host_line_boxes: per peak location the union of the components [ulx, uly, lrx, lry] that vertically coincide
 *   with the strip of half height `half` around it (textAlignPreprocessing.py:38-56, :253-276).
 */
/* The text-line peaks of a batch of pages in two calls with ONE numpy logarithm between them (numpy's float64 log is
 * its own routine on AVX-512 hosts and differs from libm's in the last bit now and then, and the prominences are
 * compared with a tolerance after a division: the logarithm stays numpy's).
 * ta_host_peak_candidates: page k's row projection proj + off[k] (len[k] int64 sums) -> smoothed + off[k]
 *   (moving_avg_filter, textAlignPreprocessing.py:147-157), its local maxima cand_idx + off[k] (cand_n[k] of them) and
 *   the ARGUMENT of each one's logarithm cand_arg + off[k] (calculate_peak_prominence :59-110).
 * ta_host_peak_select: with cand_log = log(cand_arg): peaks + off[k] (npeaks[k] rows whose prominence / the largest
 *   exceeds tol >= 0, find_peak_locations :113-144) and the white rows between neighbouring peaks rows + 2 off[k]
 *   (nrows[k], ascending; :222-232). */
int ta_host_peak_candidates(const int64_t* proj, const int64_t* off, const int32_t* len, int32_t n, int32_t filter_size,
                            double* smoothed, int32_t* cand_idx, double* cand_arg, int32_t* cand_n);
int ta_host_peak_select(const double* smoothed, const int64_t* off, const int32_t* len, int32_t n, const int32_t* cand_idx,
                        const double* cand_log, const int32_t* cand_n, double tol, int32_t* peaks, int32_t* npeaks,
                        int32_t* rows, int32_t* nrows);
int ta_host_otsu_batch(const int32_t* hist, int32_t n, int32_t* thr);
int ta_host_sharpest_rows(const int32_t* hist, const int64_t* off, const int32_t* nang, const int32_t* hs, int32_t n,
                          int32_t* best, uint8_t* any, double* var_out);
int ta_host_line_boxes(const int64_t* comps, int64_t ncomp, const int64_t* peaks, int64_t npeaks, int64_t half,
                       int64_t* out_boxes, uint8_t* out_hit);

#ifdef __cplusplus
}
#endif
#endif /* TEXT_ALIGNMENT_AMD_H */
