// sparse_host.hip -- DOGLEG_SPARSE: pattern set-up (host symbolic phase, schedule uploads,
// numeric buffers) and the K4 + K5 orchestration.  The kernels live in sparse_assemble.hip,
// sparse_factor.hip and sparse_solve.hip.
#include "sparse_internal.h"

int sparse_create(dlg_backend* b) { (void)b; return DLG_OK; }

size_t sparse_local_nnz(const dlg_backend* b) { return b->sym ? b->sym->nnz_loc : (size_t)b->nnz; }

void sparse_destroy(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y) return;
  for(void* p : Y->allocs) if(p) (void)hipFree(p);
  delete Y;
  b->sym = nullptr;
}

#define UP(field) do { DLG_CHECK(upload(Y->field, H.field)); Y->allocs.push_back(Y->field); } while(0)

int sparse_set_pattern(dlg_backend* b, const int* colptr, const int* rowidx)
{
  if(b->sym) { dlg_set_error("the sparsity pattern was already set"); return DLG_ERR_STATE; }
  if(colptr[b->M] != b->nnz)
  { dlg_set_error("Jt has %d entries but the backend was created for NJnnz = %d", colptr[b->M], b->nnz); return DLG_ERR_ARG; }
  SparseSym* Y = new (std::nothrow) SparseSym();
  if(!Y) { dlg_set_error("out of host memory"); return DLG_ERR_NOMEM; }
  b->sym = Y;
  char err[512];
  if(sym_analyze(Y->H, b->N, b->M, colptr, rowidx, b->row0, b->row1, err, sizeof(err)))
  { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  SymHost& H = Y->H;
  UP(sn_c0); UP(sn_rowptr); UP(sn_rows); UP(sn_scr); UP(lvl_sn); UP(sn_lx); UP(diagpos);
  UP(ui_t); UP(ui_col); UP(ui_nc); UP(ui_ptr); UP(usub); UP(relpos); UP(u_off); UP(usub_u); UP(fw_item); UP(mf_rec); UP(mf_dst);
  UP(uw_item); UP(uw_s0); UP(uw_s1); UP(uw_part); UP(uf_item); UP(uf_n); UP(uf_off);
  UP(oblk); UP(contrib); UP(jtx_task); UP(jtx_fin_ptr); UP(jtx_fin_blk);
  UP(asm_rho); UP(asm_pair); UP(asm_slot); UP(asm_batch); UP(asm_ctask); UP(asm_cfin);
  UP(asm_shape); UP(asm_kg); UP(asm_mtask); UP(asm_tdest); UP(asm_fin2); UP(asm_fin2_list); UP(asm_run); UP(asm_pdest);
  UP(rl_ptr); UP(rl_pos); UP(perm); UP(col_sn); UP(fw_sn); UP(fw_r0); UP(fw_r1); UP(ms_sn); UP(sn_top); UP(sn_bd_ptr); UP(sn_bd_col);
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
    const int mloc = b->row1 - b->row0;
    const int q0 = colptr[b->row0], q1 = colptr[b->row1];
    Y->nnz_loc = (size_t)(q1 - q0);
    std::vector<int> jp(mloc + 1), ji(rowidx + q0, rowidx + q1);
    for(int r = 0; r <= mloc; r++) jp[r] = colptr[b->row0 + r] - q0;
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
    DLG_CHECK(upload(Y->nv_chunk, ch)); Y->allocs.push_back(Y->nv_chunk);
  }
  auto dalloc = [&](double*& p, size_t n) -> int {
    DLG_HIP(hipMalloc(&p, sizeof(double)*(n ? n : 1))); Y->allocs.push_back(p); return DLG_OK; };
  DLG_CHECK(dalloc(Y->Lx, (size_t)H.lx_size));
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

  DLG_CHECK(sparse_factor_setup(b));
  DLG_CHECK(sparse_solve_setup(b));
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

// K4 + K5
int sparse_factorize(dlg_backend* b, int s, double lambda, int* ok)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  hipStream_t st = b->stream;
  DLG_CHECK(sparse_assemble(b, s, lambda));
  DlgProfScope pf(b, DLG_PROF_K5_FACTOR);
  if(!Y->info_armed)
  {
    // (no augmented row this time: arm the flag with a copy; the source must outlive the copy)
    static const int k_armed = 0x7fffffff;
    DLG_HIP(hipMemcpyAsync(Y->d_info, &k_armed, sizeof(int), hipMemcpyHostToDevice, st));
  }
  DLG_CHECK(sparse_factor_levels(b));
  if(pf.e) { dlg_prof_end(b, pf.id, pf.e); pf.e = nullptr; }
  // the caller's dlg_fetch_scalars(b, NSCAL) brings the flag along; sparse_factor_ok() reads it then
  if(b->defer_factor_sync) { *ok = 1; return DLG_OK; }
  DLG_HIP(hipMemcpyAsync(Y->h_info, Y->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  DLG_HIP(hipStreamSynchronize(st));
  *ok = sparse_factor_ok(b);
  return DLG_OK;
}
bool sparse_factor_ok(const dlg_backend* b) { return *b->sym->h_info == 0x7fffffff; }

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
  for(int i = 0; i < nstats && i < (int)(sizeof(v)/sizeof(v[0])); i++) stats[i] = v[i];
  if(perm_out) memcpy(perm_out, H.perm.data(), sizeof(int)*(size_t)N);
  return DLG_OK;
}
