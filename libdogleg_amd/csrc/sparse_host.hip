// sparse_host.hip -- DOGLEG_SPARSE: pattern set-up (host symbolic phase, schedule uploads,
// numeric buffers) and the K4 + K5 orchestration.  The kernels live in sparse_assemble.hip,
// sparse_factor.hip and sparse_solve.hip.
#include "sparse_internal.h"
#include <thread>
#include <mutex>
#include <memory>
#include <string>
#include <algorithm>
#include <cstring>
#include <chrono>

int sparse_create(dlg_backend* b) { (void)b; return DLG_OK; }

size_t sparse_local_nnz(const dlg_backend* b) { return b->sym ? b->sym->nnz_loc : (size_t)b->nnz; }

void sparse_destroy(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y) return;
  if(b->aux_stream) (void)hipStreamSynchronize(b->aux_stream);
  if(Y->ev_spec) (void)hipEventDestroy(Y->ev_spec);
  if(Y->ev_spec_fork) (void)hipEventDestroy(Y->ev_spec_fork);
  for(void* p : Y->allocs) if(p) (void)hipFree(p);
  delete Y;
  b->sym = nullptr;
}

#define UP(field) do { DLG_CHECK(upload(Y->field, H.field)); Y->allocs.push_back(Y->field); } while(0)

// ---- the last symbolic analysis of the process is kept ------------------------------------------
// A program that solves many problems of ONE sparsity pattern (the usual case: the same scene, new
// measurements) pays the host-side symbolic phase once: the next dlg_sparse_set_pattern with the same
// (sizes, rows of the rank, partition, pattern, schedule knobs in the environment) copies the schedules
// instead of deriving them (0.14 s -> a few ms for config #4).  The pattern itself is kept and compared
// (memcmp) behind a 64-bit hash, so a hit is exact.  DOGLEG_AMD_NO_SYM_CACHE turns it off.
namespace {
struct SymCacheEntry
{
  uint64_t key = 0;
  int N = 0, M = 0, nnz = 0, row0 = 0, row1 = 0, part_rank = 0, part_nranks = 1;
  std::vector<int> cp, ri;
  SymHost H;
};
std::mutex g_sym_mu;
std::unique_ptr<SymCacheEntry> g_sym_cache;
inline uint64_t mix64(uint64_t h, uint64_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; return h; }
uint64_t hash_ints(uint64_t h, const int* p, size_t n)
{
  size_t i = 0;
  for(; i + 1 < n; i += 2) { uint64_t v; memcpy(&v, p + i, 8); h = mix64(h, v); }
  if(i < n) h = mix64(h, (uint64_t)(uint32_t)p[i]);
  return h;
}
extern "C" char** environ;
uint64_t hash_env_knobs()
{
  // every DOGLEG_AMD_* / DLG_* variable may steer the schedules (the tests force kernels through them)
  std::vector<std::string> v;
  for(char** e = environ; e && *e; e++)
    if(!strncmp(*e, "DOGLEG_AMD_", 11) || !strncmp(*e, "DLG_", 4)) v.emplace_back(*e);
  std::sort(v.begin(), v.end());
  uint64_t h = 0x1234567ull;
  for(const std::string& t : v) for(char c : t) h = mix64(h, (uint64_t)(unsigned char)c);
  return h;
}
}

static uint64_t pattern_key(const dlg_backend* b, const int* colptr, const int* rowidx)
{
  uint64_t key = hash_env_knobs();
  const int hdr[8] = { b->N, b->M, b->nnz, b->row0, b->row1, b->part_rank, b->part_nranks, 0 };
  key = hash_ints(key, hdr, 8);
  key = hash_ints(key, colptr, (size_t)b->M + 1);
  return hash_ints(key, rowidx, (size_t)b->nnz);
}
// A backend that is used for one solve after another (the driver parks it between dogleg_optimize* calls):
// is the pattern it was set up for the one given here?  Exact: hash, then memcmp against the copy the
// symbolic cache keeps.  1: yes -- schedules, device buffers and uploads all stay; 0: no (or not known)
extern "C" int dlg_sparse_pattern_matches(dlg_backend_t* b, const int* colptr, const int* rowidx)
{
  if(!b || !b->sym || !colptr || !rowidx || b->sym->pat_key == 0) return 0;
  if(colptr[b->M] != b->nnz) return 0;
  // (no hash here: the comparison itself is the cheaper pass over the 64 MB of config #4 -- eight threads, 0.3 - 0.5 ms;
  // four took 0.5 - 1.9 ms of a 7.7 ms device-callback solve, profiles/r05_e2e.md)
  std::lock_guard<std::mutex> lk(g_sym_mu);
  const SymCacheEntry* c = g_sym_cache.get();
  if(!c || c->key != b->sym->pat_key || c->N != b->N || c->M != b->M || c->nnz != b->nnz || c->row0 != b->row0 || c->row1 != b->row1 ||
     c->part_rank != b->part_rank || c->part_nranks != b->part_nranks) return 0;
  if(memcmp(c->cp.data(), colptr, sizeof(int)*((size_t)b->M + 1))) return 0;
  const size_t n = (size_t)b->nnz;
  constexpr int NTH = 8;
  int same[NTH] = {1, 1, 1, 1, 1, 1, 1, 1};
  std::thread th[NTH];
  for(int t = 0; t < NTH; t++)
    th[t] = std::thread([&, t] { const size_t a0 = n*t/NTH, a1 = n*(t + 1)/NTH; same[t] = !memcmp(c->ri.data() + a0, rowidx + a0, sizeof(int)*(a1 - a0)); });
  for(int t = 0; t < NTH; t++) th[t].join();
  int all = 1;
  for(int t = 0; t < NTH; t++) all = all && same[t];
  return all;
}
// forget the pattern (and everything derived from it) so that another one can be set
extern "C" int dlg_sparse_drop_pattern(dlg_backend_t* b)
{
  if(!b || b->type != DLG_SPARSE) return DLG_ERR_ARG;
  if(b->stream) DLG_HIP(hipStreamSynchronize(b->stream));
  sparse_destroy(b);
  b->factor_slot = -1;
  return DLG_OK;
}
// between two solves on the same backend: nothing of the previous solve's operating points may be taken for valid
void sparse_reset(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y) return;
  Y->spec_valid = false; Y->spec_inflight = false; Y->spec_slot = -1; Y->spec_J = nullptr;
  Y->aug_rhs = nullptr; Y->spec_aug_rhs = nullptr; Y->fin_pending_rhs = nullptr; Y->fin_pending_Lx = nullptr;
  Y->info_armed = false; Y->info_clean = false;
  Y->held_Lx = nullptr;
  Y->fin_side_owed = 0; Y->fin_main = nullptr;            // (dlg_backend_reset waits for both streams first)
  // (another solve, other values: what the last one learnt about its lambda = 0 factorisations -- the look at the diagonal
  // and its synchronisation, panels left intact by a stopped attempt, a factorisation owed in part -- does not carry over)
  Y->zero_fail_seen = false; Y->intact_Lx = nullptr; Y->intact_J = nullptr; Y->intact_slot = -1;
  Y->fac_pending = false; Y->fac_J = nullptr; Y->fac_slot = -1;
}

int sparse_set_pattern(dlg_backend* b, const int* colptr, const int* rowidx)
{
  if(b->sym) { dlg_set_error("the sparsity pattern was already set"); return DLG_ERR_STATE; }
  if(colptr[b->M] != b->nnz)
  { dlg_set_error("Jt has %d entries but the backend was created for NJnnz = %d", colptr[b->M], b->nnz); return DLG_ERR_ARG; }
  SparseSym* Y = new (std::nothrow) SparseSym();
  if(!Y) { dlg_set_error("out of host memory"); return DLG_ERR_NOMEM; }
  b->sym = Y;
  const bool timing = getenv("DOGLEG_AMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if(!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "libdogleg_amd: timing:   %-32s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now; };
  char err[512];
  bool cached = false;
  const bool use_cache = getenv("DOGLEG_AMD_NO_SYM_CACHE") == nullptr;
  uint64_t key = 0;
  if(use_cache)
  {
    key = pattern_key(b, colptr, rowidx);
    Y->pat_key = key;
    std::lock_guard<std::mutex> lk(g_sym_mu);
    const SymCacheEntry* c = g_sym_cache.get();
    if(c && c->key == key && c->N == b->N && c->M == b->M && c->nnz == b->nnz && c->row0 == b->row0 && c->row1 == b->row1 &&
       c->part_rank == b->part_rank && c->part_nranks == b->part_nranks &&
       !memcmp(c->cp.data(), colptr, sizeof(int)*((size_t)b->M + 1)) && !memcmp(c->ri.data(), rowidx, sizeof(int)*(size_t)b->nnz))
    { Y->H = c->H; cached = true; }
  }
  if(!cached)
  {
    if(sym_analyze(Y->H, b->N, b->M, colptr, rowidx, b->row0, b->row1, err, sizeof(err), b->part_rank, b->part_nranks))
    { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
    if(use_cache)
    {
      std::unique_ptr<SymCacheEntry> e(new (std::nothrow) SymCacheEntry());
      if(e)
      {
        e->key = key; e->N = b->N; e->M = b->M; e->nnz = b->nnz; e->row0 = b->row0; e->row1 = b->row1;
        e->part_rank = b->part_rank; e->part_nranks = b->part_nranks;
        e->cp.assign(colptr, colptr + b->M + 1); e->ri.assign(rowidx, rowidx + b->nnz);
        e->H = Y->H;
        std::lock_guard<std::mutex> lk(g_sym_mu);
        g_sym_cache = std::move(e);
      }
    }
  }
  SymHost& H = Y->H;
  lap(cached ? "symbolic analysis (copied: same pattern as the last one)" : "symbolic analysis (host)");
  const bool partition = H.part_nranks > 1;
  if(partition) b->mloc = (int)H.part_rows.size();
  UP(sn_c0); UP(sn_rowptr); UP(sn_rows); UP(sn_scr); UP(lvl_sn); UP(sn_lx); UP(diagpos);
  UP(ui_t); UP(ui_col); UP(ui_nc); UP(ui_ptr); UP(usub); UP(relpos); UP(u_off); UP(usub_u); UP(fw_item); UP(mf_rec); UP(mf_dst);
  UP(uw_item); UP(uw_s0); UP(uw_s1); UP(uw_part); UP(uf_item); UP(uf_n); UP(uf_off);
  UP(oblk); UP(contrib); UP(jtx_task); UP(jtx_fin_ptr); UP(jtx_fin_blk);
  UP(asm_rho); UP(asm_pair); UP(asm_slot); UP(asm_batch); UP(asm_ctask); UP(asm_cfin);
  if(H.asm_jtx_ok)
  {
    UP(jf_ptr); UP(jf_ent); UP(jf_var0); UP(jf_w); UP(jf_short);
    {
      // long lists: flat records {list begin, list end, first variable, width}
      std::vector<int> rec(4*H.jf_long.size());
      for(size_t k = 0; k < H.jf_long.size(); k++)
      { const int v = H.jf_long[k]; rec[4*k] = H.jf_ptr[v]; rec[4*k+1] = H.jf_ptr[v+1]; rec[4*k+2] = H.jf_var0[v]; rec[4*k+3] = H.jf_w[v]; }
      DLG_CHECK(upload(Y->jf_long, rec)); Y->allocs.push_back(Y->jf_long);
      const size_t nl = std::max<size_t>(1, H.jf_long.size());
      DLG_HIP(hipMalloc(&Y->jf_lpart, sizeof(double)*16*JFL_SEG*nl)); Y->allocs.push_back(Y->jf_lpart);
      DLG_HIP(hipMalloc(&Y->jf_lcnt, sizeof(int)*nl)); Y->allocs.push_back(Y->jf_lcnt);
      DLG_HIP(hipMemset(Y->jf_lcnt, 0, sizeof(int)*nl));
    }
    DLG_HIP(hipMalloc(&Y->jtp, sizeof(double)*16*std::max<size_t>(1, H.asm_mtask.size()))); Y->allocs.push_back(Y->jtp);
  }
  UP(asm_shape); UP(asm_kg); UP(asm_mtask); UP(asm_tdest); UP(asm_fin2); UP(asm_fin2_list); UP(asm_run); UP(asm_pdest);
  {
    // What the END of a task stores (asm_mfma_run, TS): the persistent blocks' entries -- rows of the task's accumulator that
    // belong to a block (I, J), columns of J; the rider's diagonal block --, then the sixteen entries of the task's Jt*x
    // record, ONE entry a lane in the order that puts consecutive rows of a destination column on consecutive lanes.
    // word: bits 0-1 kind (1 block of J's panel, 2 rider's diagonal, 3 Jt*x; 0 no entry), 2-5 accumulator row, 6-9 column,
    // 10-13 slot ordinal, 14-17 row in its block, 18-25 offset of its block in a partial, 26-29 rows of its block,
    // 30 strictly above the diagonal of (J, J).  Up to two rounds of 64 (camera blocks: 81 entries); more, or fields that do not
    // fit: asm_pent_ok is false and the launch takes the kernel with the four masked rounds.
    std::vector<uint32_t> pent(128*std::max<size_t>(1, H.asm_shape.size()), 0u);
    Y->asm_pent_ok = true;
    for(size_t q = 0; q < H.asm_shape.size(); q++)
    {
      const AsmShape& sh = H.asm_shape[q];
      std::vector<uint32_t> e;
      bool fits = true;
      auto word = [&](int kind, int mm, int n) {
        if(sh.pa[mm] > 15 || sh.pnI[mm] > 15 || sh.pslot[mm] > 15) fits = false;
        return (uint32_t)kind | (uint32_t)mm << 2 | (uint32_t)n << 6 | (uint32_t)(sh.pslot[mm] & 15) << 10 | (uint32_t)(sh.pa[mm] & 15) << 14 |
               (uint32_t)sh.paccoff[mm] << 18 | (uint32_t)(sh.pnI[mm] & 15) << 26 |
               (uint32_t)((sh.pslot[mm] == sh.dslot && sh.pa[mm] < n) ? 1u : 0u) << 30; };
      for(int n = 0; n < sh.nJ; n++)
        for(int mm = 0; mm < sh.MP && mm < 16; mm++)
          if(sh.pslot[mm] != 0xFF) e.push_back(word(1, mm, n));
      for(int n = sh.nJ; n < sh.nJ + sh.nJr && n < 16; n++)
        for(int mm = 0; mm < sh.MP && mm < 16; mm++)
          if(sh.pslot[mm] != 0xFF && sh.pslot[mm] == sh.rslot) e.push_back(word(2, mm, n));
      for(int c = 0; c < 16; c++) e.push_back(3u | (uint32_t)c << 6);
      // (two rounds of 64 lanes at most; the second round's first word is non-zero exactly if there is one: an entry's kind is)
      if(!fits || e.size() > 128) { Y->asm_pent_ok = false; continue; }
      for(size_t k = 0; k < e.size(); k++) pent[128*q + k] = e[k];
    }
    DLG_CHECK(upload(Y->asm_pent, pent)); Y->allocs.push_back(Y->asm_pent);
  }
  {
    // fin on the side (sparse_assemble.hip): allowed where every block the partial-sum stages store lies in a panel
    // above level 0 (the leaf level's factor kernel runs beside them) and no LDS-kernel group has partial sums
    std::vector<std::pair<int64_t, int>> by_lx((size_t)H.nsn);
    for(int s2 = 0; s2 < H.nsn; s2++) by_lx[(size_t)s2] = { H.sn_lx[s2], s2 };
    std::sort(by_lx.begin(), by_lx.end());
    bool ok = H.asm_cfin.empty() && H.nlevels >= 2 && !getenv("DOGLEG_AMD_NO_FIN_SIDE");
    for(const AsmFin2& F : H.asm_fin2)
    {
      if(!ok) break;
      if(F.to_part) continue;
      auto it2 = std::upper_bound(by_lx.begin(), by_lx.end(), std::make_pair(F.dest, 0x7fffffff));
      if(it2 == by_lx.begin()) { ok = false; break; }
      const int s2 = (it2 - 1)->second;
      if(H.sn_level[s2] < 1) ok = false;
    }
    Y->fin_side_sched_ok = ok;
    // Partial clears (sparse_assemble.hip, clear_panels).  The panel of a merged leaf (block-diagonal members, their
    // common rows below, no children) has NO fill: a row's block under a member is (JtJ block) * (member block)^-T, zero
    // where JtJ is zero; the assembly STORES every structural entry of JtJ (persistent, transient and summed blocks
    // alike, never an addition into the panel), the leaf kernel rewrites exactly the rows below and the member blocks,
    // the augmented row is stored.  So once such a panel was cleared, everything outside the structure stays zero
    // from step to step and only the other panels -- update matrices and fill land there -- are cleared per step.
    {
      std::vector<int64_t> off, len;
      int64_t kept = 0;
      for(int s2 = 0; s2 < H.nsn; s2++)
      {
        const bool keep = H.sn_level[s2] == 0 && H.sn_bd_ptr[s2+1] > H.sn_bd_ptr[s2] && H.sn_top[s2] < 0 && H.nlevels >= 2;
        if(keep) { kept += H.sn_lx[s2+1] - H.sn_lx[s2]; continue; }
        if(!off.empty() && off.back() + len.back() == H.sn_lx[s2]) len.back() += H.sn_lx[s2+1] - H.sn_lx[s2];
        else { off.push_back(H.sn_lx[s2]); len.push_back(H.sn_lx[s2+1] - H.sn_lx[s2]); }
      }
      if(!off.empty() && off.back() + len.back() == H.lx_size) len.back() += 8;      // (+ the words behind the panels)
      else { off.push_back(H.lx_size); len.push_back(8); }
      Y->n_clr = (int)off.size();
      // (worth it where most of the buffer is kept and the ranges are few enough for one launch; LDS-kernel groups sum
      // into the panels: not with them)
      Y->clr_partial_ok = kept*2 > H.lx_size && Y->n_clr <= 65535 && H.asm_cfin.empty() && H.asm_ctask.empty() && !getenv("DOGLEG_AMD_FULL_CLEAR");
      DLG_CHECK(upload(Y->clr_off, off)); Y->allocs.push_back(Y->clr_off);
      DLG_CHECK(upload(Y->clr_len, len)); Y->allocs.push_back(Y->clr_len);
    }
    {
      // the diagonal entries of the level-0 columns (sparse_factorize: the look at the diagonal in front of a factorisation at lambda = 0)
      std::vector<int64_t> dp; std::vector<int> dc;
      for(int k = 0; k < H.N; k++) if(H.sn_level[H.col_sn[k]] == 0) { dp.push_back(H.diagpos[k]); dc.push_back(k); }
      Y->n_leaf_diag = (int)dp.size();
      if(Y->n_leaf_diag > 0)
      {
        DLG_CHECK(upload(Y->leaf_diag, dp)); Y->allocs.push_back(Y->leaf_diag);
        DLG_CHECK(upload(Y->leaf_diag_col, dc)); Y->allocs.push_back(Y->leaf_diag_col);
      }
    }
    DLG_HIP(hipMalloc(&Y->fin_flag, sizeof(int)*2)); Y->allocs.push_back(Y->fin_flag);
    DLG_HIP(hipMemsetAsync(Y->fin_flag, 0, sizeof(int)*2, b->stream));
  }
  UP(rl_ptr); UP(rl_pos); UP(perm); UP(col_sn); UP(sn_owner); UP(xl_sn); UP(fw_sn); UP(fw_r0); UP(fw_r1); UP(ms_sn); UP(sn_top); UP(sn_bd_ptr); UP(sn_bd_col);
  {
    std::vector<int64_t> ap((size_t)H.N);
    for(int k = 0; k < H.N; k++)
    {
      const int s2 = H.col_sn[k];
      const int64_t nrows = H.sn_rowptr[s2+1] - H.sn_rowptr[s2];
      ap[k] = H.sn_lx[s2] + (nrows - 1) + (int64_t)(k - H.sn_c0[s2])*nrows;
    }
    DLG_CHECK(upload(Y->augpos, ap)); Y->allocs.push_back(Y->augpos);
    // by variable (the Jt*x sums that also set the augmented rows, sparse_assemble.hip): its entry's place, and
    // whether its Jt*x is summed from a record list (then the list's workgroup stores the entry)
    std::vector<int64_t> av((size_t)H.N);
    for(int k = 0; k < H.N; k++) av[(size_t)H.perm[k]] = ap[k];
    std::vector<char> listed((size_t)H.N, 0);
    for(int pass = 0; pass < 2; pass++)
      for(int v : (pass ? H.jf_long : H.jf_short))
        for(int q = 0; q < H.jf_w[v]; q++) listed[(size_t)H.jf_var0[v] + q] = 1;
    DLG_CHECK(upload(Y->aug_of_var, av)); Y->allocs.push_back(Y->aug_of_var);
    DLG_CHECK(upload(Y->jf_listed, listed)); Y->allocs.push_back(Y->jf_listed);
  }
  {
    // Jt*x partial lists: the few long ones (a dense block that every row touches) get a big workgroup each
    std::vector<int> fs, fl;
    for(int f = 0; f + 1 < (int)H.jtx_fin_ptr.size(); f++)
      (H.jtx_fin_ptr[f+1] - H.jtx_fin_ptr[f] > 256 ? fl : fs).push_back(f);
    Y->n_fin_short = (int)fs.size(); Y->n_fin_long = (int)fl.size();
    DLG_CHECK(upload(Y->jtx_fin_short, fs)); Y->allocs.push_back(Y->jtx_fin_short);
    DLG_CHECK(upload(Y->jtx_fin_long, fl)); Y->allocs.push_back(Y->jtx_fin_long);
  }
  // rank-local pattern for the row-wise kernels
  {
    const int mloc = dlg_mloc(b);
    std::vector<int> jp(mloc + 1), ji;
    if(partition)
    {
      // the rank's rows are scattered: its local pattern is their concatenation
      jp[0] = 0;
      for(int k = 0; k < mloc; k++) { const int r = H.part_rows[k]; jp[k+1] = jp[k] + (colptr[r+1] - colptr[r]); }
      ji.resize((size_t)jp[mloc]);
      for(int k = 0; k < mloc; k++) { const int r = H.part_rows[k]; memcpy(&ji[jp[k]], rowidx + colptr[r], sizeof(int)*(size_t)(colptr[r+1] - colptr[r])); }
      Y->nnz_loc = (size_t)jp[mloc];
    }
    else
    {
      const int q0 = colptr[b->row0], q1 = colptr[b->row1];
      Y->nnz_loc = (size_t)(q1 - q0);
      ji.assign(rowidx + q0, rowidx + q1);
      for(int r = 0; r <= mloc; r++) jp[r] = colptr[b->row0 + r] - q0;
    }
    DLG_CHECK(upload(Y->Jp, jp)); Y->allocs.push_back(Y->Jp);
    DLG_CHECK(upload(Y->Ji, ji)); Y->allocs.push_back(Y->Ji);
    std::vector<int> ch; ch.push_back(0);
    for(int r = 0; r < mloc;)
    {
      int e = r;
      while(e < mloc && e - r < TPB && jp[e+1] - jp[r] <= NV_CHUNK) e++;
      if(e == r) e = r + 1;                       // one row longer than a chunk
      ch.push_back(e); r = e;
    }
    Y->n_nv_chunks = (int)ch.size() - 1;
    // flat records {first row, last row + 1, first value, values}: no dependent look-up of the row
    // pointers in front of the kernel's loads
    std::vector<int> rec(4*(size_t)Y->n_nv_chunks);
    for(int c = 0; c < Y->n_nv_chunks; c++)
    { rec[4*c] = ch[c]; rec[4*c+1] = ch[c+1]; rec[4*c+2] = jp[ch[c]]; rec[4*c+3] = jp[ch[c+1]] - jp[ch[c]]; }
    DLG_CHECK(upload(Y->nv_chunk, rec)); Y->allocs.push_back(Y->nv_chunk);
  }
  lap("schedule uploads");
  auto dalloc = [&](double*& p, size_t n) -> int {
    DLG_HIP(hipMalloc(&p, sizeof(double)*(n ? n : 1))); Y->allocs.push_back(p); return DLG_OK; };
  DLG_CHECK(dalloc(Y->Lx, (size_t)H.lx_size + 8));
  DLG_CHECK(dalloc(Y->scr, (size_t)H.scr_size));
  DLG_CHECK(dalloc(Y->ywork, (size_t)H.N));
  DLG_CHECK(dalloc(Y->upart, (size_t)H.upart_size));
  DLG_CHECK(dalloc(Y->uscr, (size_t)H.uscr_size));
  DLG_CHECK(dalloc(Y->top_scr, (size_t)H.top_size));
  DLG_CHECK(dalloc(Y->asm_part, (size_t)H.asm_part_size));
  DLG_CHECK(dalloc(Y->jtx_part, (size_t)H.jtx_nparts*8));
  // the pivot flag of the factorisation shares the backend's scalar block (last slot): it comes back
  // to the host with the scalars of the step, and the kernel that sets the augmented row re-arms it
  Y->d_info = reinterpret_cast<int*>(b->d_scal + (dlg_backend::NSCAL - 1));
  Y->h_info = reinterpret_cast<int*>(b->h_scal + (dlg_backend::NSCAL - 1));

  if(partition)
  {
    // what crosses the ranks at the cut: the panels of the replicated supernodes (kind 0: every rank
    // holds a partial sum) and the update matrices of the multifrontal region whose parent is
    // replicated and whose owner is one rank (kind 1: mine, kind 2: another rank's -- zeros from here)
    std::vector<int64_t> off, dst; std::vector<int> len, kind;
    int64_t n = 0;
    auto seg = [&](int64_t o, int64_t l, int k) {
      while(l > 0) { const int64_t c = std::min<int64_t>(l, 1 << 30); off.push_back(o); len.push_back((int)c); kind.push_back(k); dst.push_back(n); n += c; o += c; l -= c; } };
    std::vector<int> parent(H.nsn, -1);
    for(int t = 0; t < H.nsn; t++) for(int k = H.mf_cptr[t]; k < H.mf_cptr[t+1]; k++) parent[H.mf_child[k]] = t;
    for(int s2 = 0; s2 < H.nsn; s2++)
    {
      if(H.sn_owner[s2] < 0) seg(H.sn_lx[s2], H.sn_lx[s2+1] - H.sn_lx[s2], 0);
      else if(H.sn_level[s2] >= H.mf_level0 && parent[s2] >= 0 && H.sn_owner[parent[s2]] < 0)
      {
        const int64_t mb = (H.sn_rowptr[s2+1] - H.sn_rowptr[s2]) - (H.sn_c0[s2+1] - H.sn_c0[s2]);
        seg(H.u_off[s2], mb*(mb + 1)/2, H.sn_owner[s2] == H.part_rank ? 1 : 2);
      }
    }
    Y->n_red_seg = (int)off.size(); Y->red_n = (size_t)n;
    DLG_CHECK(upload(Y->red_off, off)); Y->allocs.push_back(Y->red_off);
    DLG_CHECK(upload(Y->red_len, len)); Y->allocs.push_back(Y->red_len);
    DLG_CHECK(upload(Y->red_kind, kind)); Y->allocs.push_back(Y->red_kind);
    DLG_CHECK(upload(Y->red_dst, dst)); Y->allocs.push_back(Y->red_dst);
    DLG_CHECK(dalloc(Y->red_buf, Y->red_n + 8));
    // the solution is completed by a sum over the ranks: every variable counts on exactly one rank
    std::vector<double> mask((size_t)H.N, 0.0);
    for(int k = 0; k < H.N; k++)
    {
      const int o = H.sn_owner[H.col_sn[k]];
      if(o == H.part_rank || (o < 0 && H.part_rank == 0)) mask[H.perm[k]] = 1.0;
    }
    DLG_CHECK(upload(Y->colmask, mask)); Y->allocs.push_back(Y->colmask);
  }
  lap("numeric buffers");
  DLG_CHECK(sparse_factor_setup(b));
  DLG_CHECK(sparse_solve_setup(b));
  lap("kernel set-up");
  return DLG_OK;
}

extern "C" int dlg_partition_rows(dlg_backend_t* b, int* nrows, const int** rows)
{
  if(!b || !b->sym) { dlg_set_error("no symbolic analysis yet"); return DLG_ERR_STATE; }
  SymHost& H = b->sym->H;
  if(H.part_nranks <= 1 && H.part_rows.empty())         // not partitioned: the contiguous range of the backend
    for(int r = b->row0; r < b->row1; r++) H.part_rows.push_back(r);
  if(nrows) *nrows = (int)H.part_rows.size();
  if(rows) *rows = H.part_rows.data();
  return DLG_OK;
}
// A model evaluated on the device for ALL measurement rows (dogleg_optimize_device2 on a rank of a
// multi-GPU solve): the rank's rows of x and of the values of Jt are gathered, on the device, into the
// slot's own buffers -- the order dlg_partition_rows reports -- and bound as the slot's inputs.
namespace {
__global__ void __launch_bounds__(TPB) k_gather_rows(int mloc, const int* __restrict__ rows, const int* __restrict__ src_off,
                                                     const int* __restrict__ jp_loc, const double* __restrict__ x_full,
                                                     const double* __restrict__ J_full, double* __restrict__ x_loc,
                                                     double* __restrict__ J_loc)
{
  // a wave per row: its values are a short contiguous run
  const int lane = threadIdx.x & 63;
  for(int i = blockIdx.x*(TPB/64) + (threadIdx.x >> 6); i < mloc; i += gridDim.x*(TPB/64))
  {
    const int q0 = jp_loc[i], n = jp_loc[i+1] - q0, so = src_off[i];
    if(lane == 0) x_loc[i] = x_full[rows[i]];
    for(int k = lane; k < n; k += 64) J_loc[q0 + k] = J_full[so + k];
  }
}
}
extern "C" int dlg_point_gather_device(dlg_backend_t* b, int s, const double* x_full_dev, const double* J_full_dev,
                                       const int* colptr_host)
{
  if(!b || s < 0 || s > 1 || !x_full_dev || !J_full_dev || !colptr_host) { dlg_set_error("dlg_point_gather_device: bad arguments"); return DLG_ERR_ARG; }
  if(b->type != DLG_SPARSE || !b->sym) { dlg_set_error("dlg_point_gather_device: a sparse backend with its pattern set"); return DLG_ERR_STATE; }
  SparseSym* Y = b->sym; SymHost& H = Y->H;
  int mloc = 0; const int* rows = nullptr;
  DLG_CHECK(dlg_partition_rows(b, &mloc, &rows));
  if(!Y->gat_rows)
  {
    std::vector<int> r(rows, rows + mloc), so((size_t)mloc);
    for(int i = 0; i < mloc; i++) so[i] = colptr_host[rows[i]];
    DLG_CHECK(upload(Y->gat_rows, r)); Y->allocs.push_back(Y->gat_rows);
    DLG_CHECK(upload(Y->gat_src, so)); Y->allocs.push_back(Y->gat_src);
  }
  (void)H;
  DlgSlot& S = b->slot[s];
  if(mloc > 0)
    hipLaunchKernelGGL(k_gather_rows, dim3(std::min(4096, dlg_cdiv(mloc, TPB/64))), dim3(TPB), 0, b->stream, mloc,
                       Y->gat_rows, Y->gat_src, Y->Jp, x_full_dev, J_full_dev, S.x, S.J);
  DLG_LAUNCH_CHECK();
  return dlg_point_bind_device(b, s, S.x, S.J);
}

static void partition_stats(const SymHost& H, long* stats, int nstats)
{
  long ntop = 0, nmine = 0, red = 0;
  std::vector<int> parent(H.nsn, -1);
  for(int t = 0; t < H.nsn; t++) for(int k = H.mf_cptr[t]; k < H.mf_cptr[t+1]; k++) parent[H.mf_child[k]] = t;
  for(int s2 = 0; s2 < H.nsn; s2++)
  {
    if(H.sn_owner[s2] < 0) { ntop++; red += (long)(H.sn_lx[s2+1] - H.sn_lx[s2]); }
    else
    {
      if(H.sn_owner[s2] == H.part_rank) nmine++;
      if(H.sn_level[s2] >= H.mf_level0 && parent[s2] >= 0 && H.sn_owner[parent[s2]] < 0)
      { const long mb = (H.sn_rowptr[s2+1] - H.sn_rowptr[s2]) - (H.sn_c0[s2+1] - H.sn_c0[s2]); red += mb*(mb + 1)/2; }
    }
  }
  long nnz_loc = 0;
  const long v[] = { (long)H.cut_level, ntop, nmine, (long)H.part_rows.size(), red, (long)H.lx_size, nnz_loc };
  for(int i = 0; i < nstats && i < (int)(sizeof(v)/sizeof(v[0])); i++) stats[i] = v[i];
}
extern "C" int dlg_partition_stats(dlg_backend_t* b, long* stats, int nstats)
{
  if(!b || !b->sym || !stats) { dlg_set_error("no symbolic analysis yet"); return DLG_ERR_STATE; }
  partition_stats(b->sym->H, stats, nstats);
  if(nstats > 6) stats[6] = (long)b->sym->nnz_loc;
  return DLG_OK;
}
extern "C" int dlg_sparse_partition_probe(int N, int M, const int* colptr, const int* rowidx, int rank, int nranks,
                                          long* stats, int nstats, char* row_owner)
{
  if(nranks < 1 || rank < 0 || rank >= nranks) { dlg_set_error("bad rank %d of %d", rank, nranks); return DLG_ERR_ARG; }
  SymHost H;
  char err[512];
  if(sym_analyze(H, N, M, colptr, rowidx, 0, M, err, sizeof(err), rank, nranks))
  { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  if(stats) partition_stats(H, stats, nstats);
  if(stats && nstats > 6)
  { long z = 0; for(int r : H.part_rows) z += colptr[r+1] - colptr[r]; stats[6] = nranks > 1 ? z : (long)colptr[M]; }
  if(row_owner)
  {
    memset(row_owner, nranks > 1 ? 0 : 1, (size_t)M);
    for(int r : H.part_rows) row_owner[r] = 1;
  }
  return DLG_OK;
}

extern "C" int dlg_sparse_stats(dlg_backend_t* b, long* nnz_JtJ_lower, long* nnz_L, int* n_supernodes,
                                int* n_levels, double* factor_flops)
{
  if(!b || !b->sym) { dlg_set_error("no symbolic analysis yet"); return DLG_ERR_STATE; }
  const SymHost& H = b->sym->H;
  if(nnz_JtJ_lower) *nnz_JtJ_lower = (long)H.nnz_JtJ_lower;
  if(nnz_L) *nnz_L = (long)H.nnz_L;
  if(n_supernodes) *n_supernodes = H.nsn;
  if(n_levels) *n_levels = H.nlevels;
  if(factor_flops) *factor_flops = H.factor_flops;
  return DLG_OK;
}

extern "C" int dlg_sparse_schedule(dlg_backend_t* b, int* n_levels, int* persist_level0, int* persist_items)
{
  if(!b || !b->sym) { dlg_set_error("no symbolic analysis yet"); return DLG_ERR_STATE; }
  const SparseSym* Y = b->sym; const SymHost& H = Y->H;
  const bool on = Y->pr_level0 < H.nlevels;
  if(n_levels) *n_levels = H.nlevels;
  if(persist_level0) *persist_level0 = on ? Y->pr_level0 : -1;
  if(persist_items) *persist_items = on ? H.fw_lvl_ptr[H.nlevels] - H.fw_lvl_ptr[Y->pr_level0] : 0;
  return DLG_OK;
}

// K4 + K5
namespace {
__global__ void __launch_bounds__(TPB) k_diag_look(const double* __restrict__ Lx, const int64_t* __restrict__ pos, const int* __restrict__ col, int n, int* __restrict__ info)
{
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
    if(!(Lx[pos[i]] > 0.0)) atomicMin(info, -2 - col[i]);      // (negative: found here, in front of every launch of the factorisation)
}
}
bool sparse_would_look(const dlg_backend* b, double lambda)
{
  const SparseSym* Y = b->sym;
  return Y && lambda == 0.0 && Y->zero_fail_seen && Y->n_leaf_diag > 0 && Y->H.part_nranks <= 1 && !b->sharded() && !getenv("DOGLEG_AMD_NO_DIAG_LOOK");
}
// The host has learnt that a factorisation broke down.  At lambda = 0 the next ones look at the diagonal first.  Returns
// true if THIS one was stopped by that look (a negative pivot word): no launch of it stored anything -- every kernel of the
// factorisation returns at its first look at the word --, the panels hold what the assembly left (JtJ + lambda I, the
// right-hand side rows), and the next attempt of the lambda loop takes them over (sparse_assemble) instead of assembling again.
bool sparse_note_breakdown(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y) return false;
  if(Y->cur_lambda == 0.0) Y->zero_fail_seen = true;
  Y->intact_Lx = nullptr;
  if(*Y->h_info < 0 && Y->fac_J)
  {
    Y->intact_Lx = Y->Lx; Y->intact_slot = Y->fac_slot; Y->intact_J = Y->fac_J; Y->intact_lambda = Y->cur_lambda;
    return true;
  }
  return false;
}
int sparse_factorize(dlg_backend* b, int s, double lambda, int* ok)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  hipStream_t st = b->stream;
  Y->cur_lambda = lambda;
  DLG_CHECK(sparse_assemble(b, s, lambda));
  Y->fac_slot = s; Y->fac_J = b->slot[s].Jin();
  DlgProfScope pf(b, DLG_PROF_K5_FACTOR);
  if(!Y->info_armed)
  {
    // (no augmented row this time: arm the flag with a copy; the source must outlive the copy)
    static const int k_armed = 0x7fffffff;
    DLG_HIP(hipMemcpyAsync(Y->d_info, &k_armed, sizeof(int), hipMemcpyHostToDevice, st));
  }
  Y->info_clean = false;
  Y->fac_pending = false;
  if(sparse_would_look(b, lambda))
  {
    hipLaunchKernelGGL(k_diag_look, dim3(dlg_cdiv(Y->n_leaf_diag, 4*TPB)), dim3(TPB), 0, st, (const double*)Y->Lx, Y->leaf_diag, Y->leaf_diag_col, Y->n_leaf_diag, Y->d_info);
    DLG_LAUNCH_CHECK();
    // The host asks at once (this backend HAS broken down at lambda = 0 before: a synchronisation of ~20 us against the
    // launches of a whole doomed attempt -- K5's levels, K6, K7, K8, each returning at its first look at the word: 0.4 ms on
    // config #5): doomed -> nothing else is enqueued, the caller's lambda loop goes on with panels that are the assembly's.
    DLG_HIP(hipMemcpyAsync(Y->h_info, Y->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    DLG_HIP(hipStreamSynchronize(st));
    if(*Y->h_info < 0)
    {
      if(pf.e) { dlg_prof_end(b, pf.id, pf.e); pf.e = nullptr; }
      (void)sparse_note_breakdown(b);
      b->factor_doomed = true;
      *ok = 0;
      return DLG_OK;
    }
  }
  DLG_CHECK(sparse_factor_levels(b, b->factor_ahead ? 1 : 0));
  if(pf.e) { dlg_prof_end(b, pf.id, pf.e); pf.e = nullptr; }
  // the caller's dlg_fetch_scalars(b, NSCAL) brings the flag along; sparse_factor_ok() reads it then
  if(b->defer_factor_sync) { *ok = 1; return DLG_OK; }
  DLG_HIP(hipMemcpyAsync(Y->h_info, Y->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  DLG_HIP(hipStreamSynchronize(st));
  *ok = sparse_factor_ok(b);
  return DLG_OK;
}
bool sparse_factor_pending(const dlg_backend* b) { return b->sym && b->sym->fac_pending; }
int sparse_norm2_chunks(const dlg_backend* b) { return b->sym ? b->sym->n_nv_chunks : 0; }      // workgroups (= partial sums) of K3 / K8
// the levels above the leaves of a factorisation whose leaf level was enqueued ahead (sparse_factor_levels(b, 1))
int sparse_factorize_rest(dlg_backend* b, bool* was_pending)
{
  SparseSym* Y = b->sym;
  *was_pending = Y && Y->fac_pending;
  if(!*was_pending) return DLG_OK;
  Y->fac_pending = false;
  b->prof_cont = true;
  struct Cont { dlg_backend* b; ~Cont() { b->prof_cont = false; } } cont{b};
  DlgProfScope pf(b, DLG_PROF_K5_FACTOR);
  return sparse_factor_levels(b, 2);
}
double sparse_current_lambda(const dlg_backend* b) { return b->sym ? b->sym->cur_lambda : 0.0; }
// [supernode][min, max] of the diagonal of L as the last backward solve saw it (k_solve_bwd_level: mm), or null
const double* sparse_pivot_minmax(const dlg_backend* b, int* n)
{
  *n = b->sym ? b->sym->n_diag_mm : 0;
  return (b->sym && b->sym->n_diag_mm > 0) ? b->sym->diag_mm : nullptr;
}
// A factorisation enqueued ahead of the caller's decision (backend.hip, step_prepare) takes the place of the held
// one: the held panels stay in the other buffer (sparse_assemble swaps, the buffer is cleared only behind the
// next step) and come back if the caller turns to the held factor after all (a rejected trial point).
void sparse_hold_factor(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  Y->held_Lx = Y->Lx; Y->held_aug = Y->aug_rhs; Y->held_lambda = Y->cur_lambda;
}
// the enqueued factorisation is the one the caller went on with: the displaced factor is nobody's any more
void sparse_release_held(dlg_backend* b) { if(b->sym) b->sym->held_Lx = nullptr; }
int sparse_restore_factor(dlg_backend* b, bool* restored, bool rearm)
{
  SparseSym* Y = b->sym;
  *restored = false;
  Y->fac_pending = false;
  if(!Y->held_Lx || Y->held_Lx != Y->Lx_spec) { Y->held_Lx = nullptr; return DLG_OK; }      // (assembled in place: gone)
  std::swap(Y->Lx, Y->Lx_spec);
  Y->spare_zeroed = false; Y->spare_dirty = true;                // the dropped factor: cleared behind the next step
  Y->aug_rhs = Y->held_aug; Y->cur_lambda = Y->held_lambda; Y->held_Lx = nullptr;
  static const int k_armed = 0x7fffffff;                         // (the held factor was a good one)
  if(rearm) DLG_HIP(hipMemcpyAsync(Y->d_info, &k_armed, sizeof(int), hipMemcpyHostToDevice, b->stream));
  Y->info_clean = false; Y->info_armed = false;
  *restored = true;
  return DLG_OK;
}
bool sparse_factor_ok(const dlg_backend* b) { return *b->sym->h_info == 0x7fffffff; }
// The factorisation and the solve enqueued ahead of the caller's decision (backend.hip, step_prepare) are not going to
// be used -- the trial point was rejected: their launches that have not started yet return behind their first
// barrier, exactly as they do behind a failed pivot (k_factor_level, k_update_*, k_solve_bwd_level look at the
// pivot flag; the one-launch regions still raise their flags, nothing waits for a workgroup that gave up).  The
// word is lowered from the side (the copy stream is idle here; the main stream is where those launches queue) and
// the main stream re-arms it behind them: an event orders the two writes, at the price of one barrier packet on a
// path that has just saved 0.4 ms.  What the buffers hold afterwards is what a finished factorisation leaves in
// the same places or what the assembly stored there (structural entries of the leaves' panels; everything above
// is cleared before its next use), so partial clears stay valid (tests/test_sparse_gpu.py, the reject scripts
// against DOGLEG_AMD_NO_ABANDON bit for bit).
namespace { __global__ void k_lower_word(int* word, int v) { if(threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
int sparse_abandon_enqueued(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->d_info || !b->copy_stream || !b->ev_copy) return DLG_OK;
  hipLaunchKernelGGL(k_lower_word, dim3(1), dim3(64), 0, b->copy_stream, Y->d_info, 0);
  DLG_LAUNCH_CHECK();
  DLG_HIP(hipEventRecord(b->ev_copy, b->copy_stream));
  DLG_HIP(hipStreamWaitEvent(b->stream, b->ev_copy, 0));
  hipLaunchKernelGGL(k_lower_word, dim3(1), dim3(64), 0, b->stream, Y->d_info, 0x7fffffff);      // (armed again behind what was abandoned)
  DLG_LAUNCH_CHECK();
  Y->info_clean = false; Y->info_armed = false; Y->fac_pending = false;
  return DLG_OK;
}

// host-only: run the symbolic phase on a pattern and report its statistics
// (no GPU needed; used by the CPU test-suite and by tools/)
extern "C" int dlg_sparse_symbolic_probe(int N, int M, const int* colptr, const int* rowidx, int row0,
                                         int row1, long* stats, int nstats, int* perm_out)
{
  SymHost H;
  char err[512];
  if(sym_analyze(H, N, M, colptr, rowidx, row0, row1, err, sizeof(err)))
  { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  const long v[] = { (long)H.nvb, (long)H.nsn, (long)H.nlevels, (long)H.nnz_JtJ_lower, (long)H.nnz_L,
                     (long)H.lx_size, (long)H.factor_flops, (long)H.max_panel, (long)H.asm_ctask.size(),
                     (long)H.ui_t.size(), (long)H.relpos.size(), (long)H.oblk.size(),
                     (long)H.contrib.size(), (long)H.usub.size(), (long)H.scr_size,
                     (long)H.jtx_task.size(), (long)H.asm_mtask.size(), (long)H.asm_kg.size(),
                     (long)H.asm_shape.size() };
  if(getenv("DOGLEG_AMD_SYM_DEBUG"))
    for(int l = 0; l + 1 < (int)H.uw_lvl_ptr.size(); l++)
    {
      long nu = 0, direct = 0, part = 0, subs = 0;
      for(int u = H.uw_lvl_ptr[l]; u < H.uw_lvl_ptr[l+1]; u++)
      {
        const int item = H.uw_item[u], t = H.ui_t[item];
        const long slab = (long)(H.sn_rowptr[t+1] - H.sn_rowptr[t])*H.ui_nc[item];
        nu++; subs += H.uw_s1[u] - H.uw_s0[u];
        (H.uw_part[u] < 0 ? direct : part) += slab;
      }
      if(nu) fprintf(stderr, "update level %d: %ld units, %ld sub-tasks, slabs applied directly %ld doubles, partial slabs %ld doubles, U scratch %ld doubles, two-phase %d\n",
                     l, nu, subs, direct, part, (long)H.uscr_size, (int)H.upd_syrk[l]);
    }
  for(int i = 0; i < nstats && i < (int)(sizeof(v)/sizeof(v[0])); i++) stats[i] = v[i];
  if(perm_out) memcpy(perm_out, H.perm.data(), sizeof(int)*(size_t)N);
  return DLG_OK;
}
