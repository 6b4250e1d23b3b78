// sparse_leaf_host.cpp -- host-only check of the leaf-front schedules (sparse_symbolic.h, LfLeaf): the schedule of
// every leaf is EXECUTED on the CPU the way k_leaf_front (sparse_leaf.hip) reads it -- staging table, strip tasks
// (arithmetic row patterns and lists, persistent and transient window positions, riders, split strips) -- and what
// it forms (panel rows, member blocks, direct contributions to the update matrix, Jt*x shares) is compared with the
// same sums taken straight from the pattern.  No GPU: used by the CPU test-suite (tests/test_library_cpu.py).
#include "sparse_symbolic.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

void dlg_set_error(const char* fmt, ...);          // (backend.hip)

// stats = {leaf fronts on (0/1), leaves, strip tasks, strips without a list, riders' carriers, LDS bytes}; *max_err: the
// largest |schedule's sum - direct sum| over everything a leaf forms, relative to the largest direct sum.
// Returns 0, or 1 with a message if the pattern gets no leaf fronts / a schedule is inconsistent.
extern "C" int dlg_sparse_leaf_probe(int N, int M, const int* colptr, const int* rowidx, const double* Jv, const double* x,
                                     long* stats, int nstats, double* max_err)
{
  SymHost S;
  char err[512];
  if(sym_analyze(S, N, M, colptr, rowidx, 0, M, err, sizeof(err))) { dlg_set_error("symbolic analysis: %s", err); return 1; }
  long st[6] = { (long)S.lf_ok, (long)S.lf_leaf.size(), 0, 0, 0, (long)S.lf_lds };
  double worst = 0.0, scale = 0.0;
  if(!S.lf_ok) { for(int i = 0; i < nstats && i < 6; i++) stats[i] = st[i]; if(max_err) *max_err = 0.0; dlg_set_error("no leaf fronts: %s", S.lf_why); return 1; }
  const int k0 = S.fw_lvl_ptr[0];
  for(size_t li = 0; li < S.lf_leaf.size(); li++)
  {
    const LfLeaf& L = S.lf_leaf[li];
    const FwItem& it = S.fw_item[k0 + li];
    const int w = L.w, nrows = L.nrows, mb = nrows - w, ldp = (mb + 1) & ~1, ntri = mb*(mb + 1)/2;
    if(w != it.w || nrows != it.nrows || L.lx != it.lx || L.u_off != it.u_off) { dlg_set_error("leaf %zu: record and work item differ", li); return 1; }
    const uint8_t* blob = &S.lf_blob[(size_t)L.blob];
    const int smax = S.lf_stride > 0 ? S.lf_smax : L.nslots;
    const int32_t* sv = reinterpret_cast<const int32_t*>(blob);
    const int32_t* sr = sv + smax;
    const uint16_t* sd = reinterpret_cast<const uint16_t*>(sv + 2*smax);
    const uint8_t* B = blob + L.o_lds;
    // (the same table by slot, for the form of the kernel that copies a leaf's rows into LDS in slot order)
    const int32_t* svs = reinterpret_cast<const int32_t*>(blob + (((size_t)10*smax + 3) & ~(size_t)3));
    const int32_t* srs = svs + smax;
    for(int g = 0; g < L.nslots; g++)
      if(sd[g] >= L.nslots || svs[sd[g]] != sv[g] || srs[sd[g]] != sr[g]) { dlg_set_error("leaf %zu: the by-slot table disagrees with the by-row one at %d", li, g); return 1; }
    // ---- staging
    std::vector<double> R((size_t)16*(L.nslots + 1), 0.0);
    std::vector<char> filled(L.nslots, 0);
    std::vector<int> rows_g(L.nslots), slot_of(L.nslots);
    for(int g = 0; g < L.nslots; g++)
    {
      const int len = (int)((uint32_t)sv[g] >> 28), off = sv[g] & 0xfffffff, row = sr[g], s = sd[g];
      if(s >= L.nslots || filled[s] || row < 0 || row >= M || colptr[row] != off || colptr[row+1] - colptr[row] != len || len > 15)
      { dlg_set_error("leaf %zu: staging entry %d is inconsistent", li, g); return 1; }
      filled[s] = 1; rows_g[g] = row; slot_of[g] = s;
      for(int k = 0; k < len; k++) R[(size_t)16*s + k] = Jv[off + k];
      R[(size_t)16*s + 15] = x[row];
    }
    // ---- what the kernel's LDS would hold
    std::vector<double> P((size_t)ldp*w, 0.0), Us((size_t)ntri + 2, 0.0), Dg((size_t)4*w + 2, 0.0), Sc((size_t)128*std::max(1, L.nscr), 0.0), jt((size_t)mb, 0.0);
    std::vector<char> jt_set(mb, 0);
    const LfTask* tk = reinterpret_cast<const LfTask*>(B + L.o_task);
    const LfComb* cb = reinterpret_cast<const LfComb*>(B + L.o_comb);
    const uint32_t* rbh = reinterpret_cast<const uint32_t*>(B + L.o_rbh);
    const uint8_t* fib = B + L.o_fi;
    const uint16_t* l16 = reinterpret_cast<const uint16_t*>(B);
    auto tri = [&](int j) { return j*mb - j*(j - 1)/2 - j; };
    auto put = [&](double acc, int kind, int pdk, int k, int j, const LfTask& T) {
      if(j >= T.wj || pdk == 0xFF || pdk == 0xFE) return;
      if(kind == 0)
      {
        if(pdk == 0xFD) { const int c = k - T.kj; if(c >= j) Dg[(size_t)(T.col0 + c)*4 + j] = acc; }
        else P[(size_t)pdk + (size_t)(T.col0 + j)*ldp] = acc;
      }
      else
      {
        const int c = T.col0 + j;
        if(pdk == 0xFC) { jt[c] = acc; jt_set[c] = 1; }
        else if(pdk >= c) Us[(size_t)tri(c) + pdk] = acc;
      } };
    for(int t = 0; t < L.ntask; t++)
    {
      const LfTask& T = tk[t];
      st[2]++; if(T.flags & 1) st[3]++; if(T.rwj) st[4]++;
      // the task's rows
      std::vector<int> slots;
      if(T.flags & 1) { for(int o = 0; o < T.a_nout; o++) for(int r = 0; r < T.a_nin; r++) slots.push_back(T.a_slot0 + o*T.a_stride + r); }
      else for(int i = 0; i < T.nprow; i++) slots.push_back(l16[T.plist + i]);
      if((int)slots.size() != T.nprow) { dlg_set_error("leaf %zu task %d: %zu rows in the pattern, %d in the record", li, t, slots.size(), (int)T.nprow); return 1; }
      for(int sl : slots) if(sl < 0 || sl >= L.nslots) { dlg_set_error("leaf %zu task %d: a row outside the staged ones", li, t); return 1; }
      // persistent positions (and the rider's columns)
      for(int k = 0; k < 16; k++)
        for(int j = 0; j < T.wj + T.rwj; j++)
        {
          const int kb = (j < T.wj) ? T.kj + j : T.rkj + (j - T.wj);
          double acc = 0.0;
          for(int sl : slots) acc += R[(size_t)16*sl + k]*R[(size_t)16*sl + kb];
          if(j < T.wj)
          {
            if(T.kind == 2) { if(j < 8) Sc[(size_t)T.scr*128 + k*8 + j] = acc; }
            else put(acc, T.kind, T.pd[k], k, j, T);
          }
          else Sc[(size_t)T.scr*128 + k*8 + (j - T.wj)] = acc;
        }
      // transient positions
      const int nT = T.pd[31];
      for(int q = 0; q < T.ntrb && nT > 0; q++)
      {
        const int rbi = (T.flags & 2) ? T.tlist + q : l16[T.tlist + q];
        if(rbi < 0 || rbi >= L.nrb) { dlg_set_error("leaf %zu task %d: row-block out of range", li, t); return 1; }
        int sl = rbh[rbi] & 0xFFFF, h = (rbh[rbi] >> 16) & 0xFF;
        if(T.hh > 0 && (sl != T.a_slot0 + q*T.hh || h != T.hh)) { dlg_set_error("leaf %zu task %d: row-block %d is not where the pattern says", li, t, q); return 1; }
        for(int tq = 0; tq < nT; tq++)
          for(int j = 0; j < T.wj; j++)
          {
            const int kT = T.pd[16 + tq], f = fib[rbi*16 + kT];
            double acc = 0.0;
            for(int r = 0; r < h; r++) acc += R[(size_t)16*(sl + r) + kT]*R[(size_t)16*(sl + r) + T.kj + j];
            if(f >= mb) { dlg_set_error("leaf %zu task %d: transient destination outside the panel", li, t); return 1; }
            P[(size_t)f + (size_t)(T.col0 + j)*ldp] = acc;
          }
      }
    }
    for(int c = 0; c < L.ncomb; c++)
    {
      const LfComb& C = cb[c];
      const LfTask& T = tk[C.task];
      for(int k = 0; k < 16; k++)
        for(int j = 0; j < T.wj; j++)
        {
          double acc = 0.0;
          for(int q = 0; q < C.nscr; q++) acc += Sc[(size_t)(C.scr0 + q)*128 + k*8 + std::min(j, 7)];
          put(acc, 1, T.pd[k], k, j, T);
        }
    }
    // ---- the same sums straight from the pattern: front index of a variable = position in the supernode's row list
    const int* srows = &S.sn_rows[S.sn_rowptr[it.s]];
    std::map<int, int> fidx;                                   // elimination position -> front index
    for(int i = 0; i < nrows - 1; i++) fidx[srows[i]] = i;
    std::map<std::pair<int, int>, double> ref;                 // (front row >= front column) -> sum; front row nrows - 1: the right-hand side
    for(int g = 0; g < L.nslots; g++)
    {
      const int row = rows_g[g];
      std::vector<std::pair<int, double>> e;
      for(int q = colptr[row]; q < colptr[row+1]; q++)
      {
        auto f = fidx.find(S.iperm[rowidx[q]]);
        if(f == fidx.end()) { dlg_set_error("leaf %zu: row %d has a variable outside the front", li, row); return 1; }
        e.push_back({f->second, Jv[q]});
      }
      e.push_back({nrows - 1, x[row]});
      for(auto& a : e) for(auto& b : e) if(a.first >= b.first && b.first < nrows - 1) ref[{a.first, b.first}] += a.second*b.second;
    }
    for(auto& kv : ref) scale = std::max(scale, std::fabs(kv.second));
    // every reference entry must be where the kernel would find it; everything else the schedule wrote must be zero in the reference
    std::vector<double> Pz(P), Uz(Us), Dz(Dg);
    for(auto& kv : ref)
    {
      const int i = kv.first.first, j = kv.first.second;
      double got;
      if(j < w)
      {
        if(i < w) { if(i/L.bdw != j/L.bdw) { dlg_set_error("leaf %zu: members %d and %d couple", li, i, j); return 1; } got = Dg[(size_t)i*4 + (j % L.bdw)]; Dz[(size_t)i*4 + (j % L.bdw)] = 0.0; }
        else { got = P[(size_t)(i - w) + (size_t)j*ldp]; Pz[(size_t)(i - w) + (size_t)j*ldp] = 0.0; }
      }
      else if(i == nrows - 1) { got = jt_set[j - w] ? jt[j - w] : 0.0; }
      else { got = Us[(size_t)tri(j - w) + (i - w)]; Uz[(size_t)tri(j - w) + (i - w)] = 0.0; }
      worst = std::max(worst, std::fabs(got - kv.second));
    }
    for(double v : Pz) worst = std::max(worst, std::fabs(v));
    for(size_t e = 0; e < (size_t)ntri; e++) worst = std::max(worst, std::fabs(Uz[e]));
    for(size_t e = 0; e < (size_t)4*w; e++) worst = std::max(worst, std::fabs(Dz[e]));
  }
  for(int i = 0; i < nstats && i < 6; i++) stats[i] = st[i];
  if(max_err) *max_err = scale > 0.0 ? worst/scale : worst;
  return 0;
}
