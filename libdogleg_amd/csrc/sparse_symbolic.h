// sparse_symbolic.h -- host-side symbolic phase of the sparse path.  Replaces
// cholmod_analyze (reference call site dogleg.c:650-654): run once per solve on
// the constant sparsity pattern of Jt (dogleg.c:648-649).
//
// Output = everything the numeric kernels need, as flat arrays that are
// uploaded verbatim:
//   * a block-sparse description of J: "var-blocks" (runs of state variables
//     that always appear together) and "row-blocks" (runs of measurement rows
//     with identical pattern), so that addresses of Jacobian values are computed
//     from a few integers per row-block instead of per-entry index lists;
//   * a fill-reducing, parallelism-exposing elimination order of the var-blocks
//     (nested dissection by BFS level structures, leaves ordered by degree,
//     dense blocks last);
//   * supernodes (block columns with nested structure merged), their dense
//     column-major panels, the supernodal elimination-tree levels;
//   * every panel carries one extra ("augmented") last row that holds the
//     right-hand side Jt_x of its columns: the factorisation then produces
//     y = L^-1 P Jt_x as a by-product (Cholesky of [A b; b' .]), so the
//     Gauss-Newton solve needs only the backward substitution;
//   * gather lists: which row-blocks contribute to which JtJ block (assembly,
//     Jt*x), which descendant panels update which target panel columns
//     (factorisation), which scratch entries feed which row (forward solve).
#pragma once
#include <cstdint>
#include <vector>

// one contribution of a row-block to an output block: 16 bytes, read as int4
struct SymContrib
{
  int32_t  base;     // position of the row-block's first value in the (rank-local) value array
  int32_t  r0;       // first measurement row of the row-block, rank-local
  uint16_t len;      // entries per row
  uint8_t  nrows;    // rows in the row-block
  uint8_t  pad;
  uint16_t offI;     // offset of var-block I inside a row
  uint16_t offJ;     // offset of var-block J inside a row
};
static_assert(sizeof(SymContrib) == 16, "SymContrib must be 16 bytes");

// a JtJ output block (I,J), pos(I) >= pos(J) in elimination order
struct SymOutBlock
{
  int64_t dest;      // offset in Lx of element (first row of I, first col of J)
  int32_t ld;        // leading dimension of the target panel
  int32_t var0;      // first ORIGINAL variable index of I (used by Jt*x on diagonal blocks)
  uint8_t nI, nJ, diag, pad;
  int32_t pad2;
};
static_assert(sizeof(SymOutBlock) == 24, "SymOutBlock layout");

// one factor-update sub-task: the rows [ka, nrows_d) of source panel d update
// one var-block of columns of a target panel
struct SymSub
{
  int64_t src;       // Lx offset of source panel element (row ka, col 0)
  int32_t rel;       // index into relpos of the entry for row ka
  int32_t nrows_d;   // leading dimension of the source panel
  int32_t wd;        // columns of the source panel
  int32_t m;         // nrows_d - ka
};
static_assert(sizeof(SymSub) == 24, "SymSub layout");

// ---- JtJ assembly schedule (column-block centric, see sparse_symbolic.cpp 9b)
struct AsmRho                // one row-block inside a task
{
  int32_t  base;             // first value of the row-block in the (rank-local) value array
  int32_t  pair0;            // first pair record; the next AsmRho's pair0 closes the list
  uint16_t len;              // entries per row
  uint16_t offJ;             // offset of the task's column block inside a row
  uint16_t stage_off;        // offset in the LDS staging buffer; 0xFFFF: too big, read from HBM
  uint8_t  nrows, pad;
};
static_assert(sizeof(AsmRho) == 16, "AsmRho layout");
struct AsmPair { uint16_t offI; uint16_t acc_nI; };   // acc_nI = accumulator offset | (nI-1) << 12
struct AsmSlot { int64_t dest; int32_t ld; uint16_t accoff; uint8_t nI, diag; };
static_assert(sizeof(AsmSlot) == 16, "AsmSlot layout");
struct AsmBatch { int32_t rho0, rho1; };
struct AsmTask
{
  int32_t  batch0, batch1, slot0;
  uint16_t nslots; uint8_t nJ, pad;
  int32_t  acc_size, pad2;
  int64_t  part;             // offset into the partial buffer, or -1: write the panels directly
};
static_assert(sizeof(AsmTask) == 32, "AsmTask layout");
struct AsmFin { int32_t slot0, nslots, acc_size, nparts; int64_t part0; int32_t nJ, pad; };
static_assert(sizeof(AsmFin) == 32, "AsmFin layout");

// ---- MFMA assembly path (column blocks whose row-blocks all have the same layout)
// A "k-group" is 4 Jacobian rows fed to one v_mfma_f64_16x16x4_f64.  The blocks
// (I,J) of a task's column J are of two kinds: PERSISTENT (the same I in every
// row-block: the diagonal block, dense columns) accumulate in the MFMA
// accumulator over the whole task; TRANSIENT (a different I in every row-block,
// each receiving exactly one contribution) are stored after each k-group.
struct AsmShape
{
  int16_t pcol[16], tcol[16];   // offset inside a row of persistent / transient output row m, -1: none
  uint8_t pslot[16], pa[16];    // persistent row m: slot ordinal, row inside the block
  uint8_t tj[16], ta[16];       // transient row m: transient ordinal, row inside the block
  uint16_t offJ;                // offset of the column block inside a row
  uint8_t nJ, MP, MT, nT, smax, pad;
  // "rider": a dense column block J' (its only output is its own diagonal block, every row-block
  // lists it last) does not get tasks of its own; the tasks of another column block that read
  // the same rows carry J' as nJr extra B columns and emit a partial of (J',J')
  uint16_t offR;                // offset of the rider block inside a row
  uint8_t nJr, rslot;           // rider width (0: none), persistent slot ordinal of the rider's rows
  // the kernel stages columns [col0, col0 + ncopy) of every row in LDS; pcol/tcol/offJ/offR are
  // relative to col0
  uint16_t col0;
  uint8_t ncopy, dslot;         // dslot: persistent slot ordinal of the diagonal block (J,J)
  uint8_t paccoff[16], pnI[16]; // persistent row m: offset of its block column 0 entry in a partial, rows of its block
};
static_assert(sizeof(AsmShape) == 176, "AsmShape layout");
struct AsmKG
{
  int32_t  base[4];             // first value of each of the 4 rows, -1: no row
  int32_t  tq;                  // first entry in asm_tdest of row-block slot 0 (slot s: + s*nT)
  uint32_t meta;                // bits 0-7: row-block slot of each row (2 bits each); 8-10: #slots; 11: store transients; 13: td[] holds the destinations
  int32_t  xr[4];               // the rows themselves (index into the rank's x), 0 where there is none: the
                                // kernel can form Jt*x beside JtJ (it holds every J(row, column of J) anyway)
  int32_t  td[2];               // meta bit 13: the k-group has at most two transient destinations and they are
                                // HERE (no dependent load of asm_tdest between the record and the stores)
};
static_assert(sizeof(AsmKG) == 48, "AsmKG layout");
constexpr int ASM_KG_DW = 12;   // dwords per record
constexpr int ASM_KG_ALIGN = 4; // a run's k-groups are padded to a multiple of this (the kernel's unroll, ASM_U)
struct AsmMTask
{
  int32_t kg0, kg1, slot0, shape;
  int32_t ld;
  int32_t pq;                   // first entry in asm_pdest: row offset in J's panel of every persistent block
  int64_t panel;                // Lx offset of (row 0, first column of J) of J's panel
  int64_t part;                 // offset into the partial buffer, or -1: write the panels directly
  int64_t rpart;                // partial of the rider's diagonal block (shape.nJr > 0)
  int32_t jvar;                 // Jt*x beside JtJ: first variable of J if this is the ONLY task of J (the kernel
  int32_t pad;                  // writes (Jt x)[J] itself), else -1 (its record is summed by k_jtx_fin2_*)
};
static_assert(sizeof(AsmMTask) == 56, "AsmMTask layout");
constexpr int ASM_MTASK_DW = 14;
// a wave's work: consecutive tasks of one shape; their k-groups are contiguous
struct AsmRun { int32_t task0, task1, kg0, kg1; };
// a persistent block written by several tasks: sum of the listed partials, in list order
// to_part: an intermediate sum of a long list, written densely to the partial buffer at dest
struct AsmFin2 { int64_t dest; int32_t ld, list0, nlist; uint8_t nI, nJ, diag, to_part; };
static_assert(sizeof(AsmFin2) == 24, "AsmFin2 layout");

// a wave-task: contributions [c0,c1) of block blk; part >= 0: write the partial
// into slot `part` of the partial buffer instead of the destination
struct SymTask { int32_t blk, c0, c1, part, var0, nI; };   // var0 / nI: copied from the out-block (one dependent load less)

// LDS bytes a factor workgroup may use: (almost) all 160 KB of a CU, the rest is its static LDS
constexpr int SYM_FAC_LDS_BUDGET = 163840 - 3584 - 8704;      // (static LDS of the factor kernel: its own words + panel_factor_b16's hand-off buffers)

// Where the update matrix W (mb x mb, packed lower triangle) of an unsliced supernode lives while its
// factor workgroup runs: behind the panel in LDS if both fit; else its last columns move into the
// unused strict upper triangle of the panel's top block -- column jw of W (length mb - jw) fills
// exactly the mb - jw free slots at the head of panel column mb - jw -- which is possible for the
// columns jw >= mb - w + 1.  Returns the first column that lives up there (mb: none, everything is
// behind the panel; -1: W does not fit either way and stays in HBM).
inline int sym_w_split(long w, long nrows)
{
  const long mb = nrows - w, ldp = (nrows + 1) & ~1L, ntri = mb*(mb + 1)/2, room = SYM_FAC_LDS_BUDGET/8 - ldp*w - 1;
  if(ntri <= room) return (int)mb;
  const long jsp = mb - w + 1 > 0 ? mb - w + 1 : 0;
  if(jsp*mb - jsp*(jsp - 1)/2 <= room) return (int)jsp;
  return -1;
}
inline long sym_w_linear(long mb, long jsp) { return jsp*mb - jsp*(jsp - 1)/2; }    // doubles behind the panel

// one factor work item (supernode, slice of its below rows)
struct FwItem
{
  int s, r0, r1, w;            // supernode, slice [r0, r1) of the below rows, width
  int nrows, col0;             // rows of the panel, first column (elimination position)
  int bd0, nbd;                // block-diagonal top: members in sn_bd_col[bd0 .. bd0 + nbd)
  int64_t lx, top, u_off;      // panel offset, top-block copy (or -1), update-matrix offset (or -1)
  int ch0, nch;                // multifrontal children: mf_rec[ch0 .. ch0 + nch)
  int bdw, jsp;                // block-diagonal top: the common width of the members, 0 if they differ; sym_w_split of the supernode
  // one-launch region of the factorisation (sparse_factor_setup): a supernode may be factored by several
  // workgroups ("replicas": identical arithmetic on the whole panel), each of which forms and hands over
  // the 16-column tile columns [tj0, tj1) of the update matrix; the LAST replica stores the panel, once the others have read it (rsv2 = number of replicas)
  int rep, tj0, tj1, pad;      // pad: the level (profile build's dump)
  // a replica that keeps only ITS columns of the update matrix in LDS (sliced != 0): the packed entries
  // [eA, eB) -- columns [16 tj0, 16 tj1) -- sit behind the panel, everything else of the children's update
  // matrices that is not a panel entry is dropped by its destination lists.  An update matrix that does
  // not fit LDS whole (a 66-column separator with 139 rows below: 106 KB of panel + 76 KB) fits in slices.
  int sliced, eA, eB, rsv2;
};
// one child of a supernode of the multifrontal region: its update matrix and, entry by entry
// (packed order, padded to a multiple of 1024 with a scratch slot), where each entry goes
struct MfChild { int64_t u_off, dst_off; int npad, rsv; };     // rsv: the child's work item (index into fw_item), -1 if it has none on this rank;
                                                               // in the one-launch region's copy: first workgroup of the child | its replicas << 20, -1: not in the launch

struct SymHost
{
  int N = 0, M = 0, nnz = 0;
  int row0 = 0, row1 = 0;            // local rows (contiguous row sharding)
  // subtree partition (multi-GPU): owner rank of every supernode, -1 = above the cut (replicated on
  // every rank); the supernodes this rank works on by level (xl_*: all of them on a single rank);
  // the measurement rows this rank holds, in the order of its local x / J value arrays
  int part_rank = 0, part_nranks = 1, cut_level = -1;
  std::vector<int> sn_owner, xl_ptr, xl_sn, part_rows;
  // ---- blocks
  int nvb = 0;
  std::vector<int> vb_start;         // [nvb+1] original variable index
  // ---- ordering
  std::vector<int> perm;             // [N] perm[k] = original variable at elimination position k
  std::vector<int> iperm;            // [N]
  // ---- supernodes
  int nsn = 0;
  std::vector<int>     sn_c0;        // [nsn+1] first column (elimination position)
  std::vector<int>     sn_rowptr;    // [nsn+1] into sn_rows
  std::vector<int>     sn_rows;      // row structure (elimination positions, ascending; first w = own columns)
  std::vector<int64_t> sn_lx;        // [nsn+1] panel offsets into Lx (column-major, ld = nrows)
  std::vector<int>     sn_scr;       // [nsn+1] offsets into the solve scratch (below rows only)
  std::vector<int>     sn_level;     // [nsn]
  int nlevels = 0;
  std::vector<int>     lvl_ptr;      // [nlevels+1]
  std::vector<int>     lvl_sn;       // [nsn] supernodes sorted by level
  std::vector<int>     sn_bd_ptr, sn_bd_col;   // block-diagonal-top supernodes: first column of each member block
  std::vector<int>     fw_lvl_ptr, fw_sn, fw_r0, fw_r1;   // factor work items: supernode + slice of its below rows
  std::vector<int64_t> sn_top;       // [nsn] offset of the top-block copy in top_scr for multi-slice supernodes, else -1
  std::vector<int>     ms_sn;        // the multi-slice supernodes
  int64_t top_size = 0;
  std::vector<int64_t> diagpos;      // [N] Lx offset of the diagonal entry of column k
  std::vector<int>     col_sn;       // [N] supernode of column k
  int64_t lx_size = 0;
  int     scr_size = 0;
  int     max_panel = 0;             // max nrows*w over supernodes
  // ---- factor update schedule (per source level)
  std::vector<int> ui_lvl_ptr;       // [nlevels+1] into items
  std::vector<int> ui_t, ui_col, ui_nc, ui_ptr;   // items: target, first local col, #cols, subtask range [ni+1]
  std::vector<SymSub> usub;          // sub-tasks, grouped by item
  std::vector<int> relpos;           // row positions in the target panel
  // work units = chunks of an item's sub-tasks (long lists, e.g. a dense last block, are split)
  std::vector<int> uw_lvl_ptr;       // [nlevels+1] into units
  std::vector<int> uw_item, uw_s0, uw_s1;
  std::vector<int64_t> uw_part;      // offset into the partial-slab buffer, or -1: apply directly
  std::vector<int> uf_lvl_ptr;       // [nlevels+1] into finalize entries
  std::vector<int> uf_item, uf_n;    // item, number of partial slabs
  std::vector<int64_t> uf_off;       // offset of the first partial slab
  int64_t upart_size = 0;
  // two-phase update of a level (many small sources, e.g. the point leaves of a bundle
  // adjustment): phase 1 forms U_d = B_d B_d' (B_d: rows below the diagonal block) of every
  // source once, phase 2 gathers the column blocks of the U_d into the targets
  std::vector<char>    upd_syrk;     // [nlevels] the level uses the two-phase update
  std::vector<int64_t> u_off;        // [nsn] offset of U_d in the scratch (two-phase levels only)
  std::vector<int64_t> usub_u;       // [#sub-tasks] offset of the sub-task's first element in the scratch
  int64_t uscr_size = 0;
  // multifrontal top of the tree (levels >= mf_level0; == nlevels: none): a supernode of the
  // region does not update its ancestors one by one, it hands its whole update matrix
  // U = (children's leftovers) + B B' (packed lower triangle, column-major) to its parent, whose
  // factor workgroup adds it into its panel / its own U before factorising (k_factor_level)
  int mf_level0 = 0;
  std::vector<int> mf_cptr, mf_child;   // [nsn+1], children inside the region (ascending)
  std::vector<int> sn_prel;             // [nsn] relpos offset of "below rows of s -> rows of parent(s)", -1: root
  // flat records for k_factor_level: everything a workgroup needs in one load (a dependent
  // global load costs microseconds at the top of the tree, where little else is going on)
  std::vector<FwItem>  fw_item;         // per factor work item, same order as fw_sn
  std::vector<MfChild> mf_rec;          // children of the multifrontal region, grouped by parent
  std::vector<uint16_t> mf_dst;         // destination of every entry of every child: element offset in the parent's
                                        // LDS panel / update matrix; bit 15: offset into an update matrix kept in HBM
  // ---- assembly / Jt*x
  std::vector<SymOutBlock> oblk;     // diagonal block of every var-block (Jt*x and lambda use them)
  std::vector<SymContrib>  contrib;
  std::vector<SymTask>     jtx_task; // Jt*x wave-tasks (diagonal blocks only)
  std::vector<int> jtx_fin_ptr, jtx_fin_blk;
  int jtx_nparts = 0;
  bool jtx_covers_all = false;       // every var-block has a (local) contribution: k_jtx writes all of Jt_x
  std::vector<AsmRho>   asm_rho;
  std::vector<AsmPair>  asm_pair;
  std::vector<AsmSlot>  asm_slot;
  std::vector<AsmBatch> asm_batch;
  std::vector<AsmTask>  asm_ctask;
  std::vector<AsmFin>   asm_cfin;
  std::vector<AsmShape> asm_shape;
  std::vector<AsmKG>    asm_kg;
  std::vector<AsmMTask> asm_mtask;
  std::vector<AsmFin2>  asm_fin2;     // grouped in stages (long lists are summed hierarchically)
  std::vector<int64_t>  asm_fin2_list;
  int asm_lds_len = 0;                // LDS row stride (doubles) of the MFMA assembly kernel
  bool asm_td_inline = false;         // every k-group with transient blocks carries their destinations in its record (AsmKG::td) and stores them (meta bit 11)
  bool asm_ts_off = false;          // DOGLEG_AMD_ASM_MFMA=2: the MFMA assembly with the four masked transient stores a k-group (the form that serves shapes whose entries do not fit 64 lanes)
  // Jt*x out of the assembly kernel: every MFMA task leaves a 16-double record (lanes 0..nJ-1: its own
  // column block, nJ..: the rider it carries); var-block v sums the records jf_ent[jf_ptr[v] .. jf_ptr[v+1])
  // (entry = 16*task + first lane) in that order.  asm_jtx_ok: every var-block with rows is covered.
  bool asm_jtx_ok = false;
  std::vector<int> jf_ptr, jf_ent, jf_var0, jf_w, jf_short, jf_long;
  std::vector<int>      fin2_stage;   // per stage: first entry, #entries with <= 64 partials, #entries with more
  std::vector<AsmRun>   asm_run;
  std::vector<int>      asm_pdest;
  std::vector<int>      asm_tdest;   // row offset in J's panel of every (row-block, transient ordinal)
  int64_t asm_part_size = 0;
  // ---- forward-solve gather lists
  std::vector<int> rl_ptr;           // [N+1]
  std::vector<int> rl_pos;           // scratch positions feeding row k
  // ---- statistics
  int64_t nnz_JtJ_lower = 0, nnz_L = 0;
  double  factor_flops = 0;
};

// Build everything.  colptr/rowidx: FULL pattern of Jt (CSC, Nstate x Nmeas).
// Returns 0 on success; on failure returns nonzero and fills err.
// part_nranks > 1: subtree partition -- the rank's rows are chosen by the analysis (S.part_rows),
// row0 / row1 are ignored.
int sym_analyze(SymHost& S, int N, int M, const int* colptr, const int* rowidx, int row0, int row1,
                char* err, int errlen, int part_rank = 0, int part_nranks = 1);
