// sparse_symbolic.cpp -- see sparse_symbolic.h.  Pure host code (no HIP).
#include "sparse_symbolic.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <map>
#include <cmath>
#include <numeric>
#include <queue>
#include <array>

namespace {

constexpr int VB_MAX   = 8;       // max variables per var-block (<= 64 lanes per 8x8 output block)
constexpr int RB_MAX   = 8;       // max rows per row-block
constexpr int CH_JTX   = 128;     // contributions per Jt*x wave-task (long lists are latency chains: keep them short)
constexpr int MAXCH_JTX = 2048;
constexpr int PANEL_CAP = 16384;  // doubles: supernode panels up to this size are factored in LDS
constexpr int SN_WMAX  = 256;     // max supernode width

int env_int(const char* name, int dflt)
{
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
// doubles: a row slice (top block + its rows) of a larger panel; as much of the 160 KB LDS as a
// single workgroup can get, so that fewer slices repeat the factorisation of the top block
static int slice_cap() { return env_int("DOGLEG_AMD_SLICE_CAP", SYM_FAC_LDS_BUDGET/8 - 40); }

struct RowBlock { int r0, nrows, len, base, vptr, nvb; bool local; int lbase, lr0; };   // lbase / lr0: first value / first row in the rank-local arrays

// ------------------------------------------------------------------ graph ---
struct Graph
{
  int n = 0;
  std::vector<int> ptr, adj;      // CSR, no self loops, sorted
  std::vector<int> w;             // node weights (variables per block)
  std::vector<long> wdeg;         // weighted degree
};

// ---------------------------------------------------- nested dissection -----
struct Orderer
{
  const Graph& G;
  std::vector<char> removed;      // dense nodes and already-ordered separators
  std::vector<int>  mark;         // generic stamp array
  int stamp = 0;
  std::vector<int>  out;          // resulting order (node ids)
  int leaf_w;

  explicit Orderer(const Graph& g) : G(g), removed(g.n, 0), mark(g.n, 0)
  { leaf_w = env_int("DOGLEG_AMD_ND_LEAF", 300); }

  // BFS over `in_set`-stamped nodes from root; fills levels; returns eccentricity
  int bfs(int root, int setstamp, std::vector<int>& order, std::vector<int>& lvl_start,
          std::vector<int>& dist)
  {
    order.clear(); lvl_start.clear();
    ++stamp;
    const int visited = stamp;
    order.push_back(root); vis[root] = visited; dist[root] = 0;
    lvl_start.push_back(0);
    size_t head = 0;
    int cur = 0;
    while(head < order.size())
    {
      const int u = order[head];
      if(dist[u] != cur) { cur = dist[u]; lvl_start.push_back((int)head); }
      head++;
      for(int e = G.ptr[u]; e < G.ptr[u+1]; e++)
      {
        const int v = G.adj[e];
        if(inset[v] != setstamp || vis[v] == visited) continue;
        vis[v] = visited; dist[v] = dist[u] + 1; order.push_back(v);
      }
    }
    lvl_start.push_back((int)order.size());
    return cur;
  }
  std::vector<int> inset, vis;

  // exact minimum degree on a small induced subgraph (leaf of the dissection)
  void order_leaf(const std::vector<int>& nodes)
  {
    const int n = (int)nodes.size();
    if(n > 4000)
    {
      std::vector<int> s(nodes);
      std::sort(s.begin(), s.end(), [&](int a, int b) {
        if(G.wdeg[a] != G.wdeg[b]) return G.wdeg[a] < G.wdeg[b];
        return a < b; });
      out.insert(out.end(), s.begin(), s.end());
      return;
    }
    // local ids
    ++stamp;
    std::vector<int> lid_of;           // via mark: store local id in a side map
    static thread_local std::vector<int> loc;
    if((int)loc.size() < G.n) loc.assign(G.n, -1);
    for(int i = 0; i < n; i++) loc[nodes[i]] = i;
    std::vector<std::vector<int>> adj(n);
    std::vector<long> ext(n, 0);       // weight of neighbours outside the leaf (static)
    for(int i = 0; i < n; i++)
    {
      const int u = nodes[i];
      for(int e = G.ptr[u]; e < G.ptr[u+1]; e++)
      {
        const int v = G.adj[e];
        if(loc[v] >= 0 && inset[v] == inset[u] && loc[v] < n && nodes[loc[v]] == v) adj[i].push_back(loc[v]);
        else ext[i] += G.w[v];
      }
      std::sort(adj[i].begin(), adj[i].end());
    }
    std::vector<char> gone(n, 0);
    std::vector<long> deg(n);
    auto wdeg = [&](int i) { long d = ext[i]; for(int v : adj[i]) d += G.w[nodes[v]]; return d; };
    for(int i = 0; i < n; i++) deg[i] = wdeg(i);
    std::vector<int> tmp;
    for(int it = 0; it < n; it++)
    {
      int best = -1;
      for(int i = 0; i < n; i++)
        if(!gone[i] && (best < 0 || deg[i] < deg[best] || (deg[i] == deg[best] && nodes[i] < nodes[best]))) best = i;
      gone[best] = 1;
      out.push_back(nodes[best]);
      // neighbours of best become a clique
      const std::vector<int>& nb = adj[best];
      for(int v : nb)
      {
        tmp.clear();
        std::set_union(adj[v].begin(), adj[v].end(), nb.begin(), nb.end(), std::back_inserter(tmp));
        // drop v itself and best
        std::vector<int> nv; nv.reserve(tmp.size());
        for(int x : tmp) if(x != v && x != best) nv.push_back(x);
        adj[v].swap(nv);
      }
      // the eliminated node's external weight is inherited by its neighbours' fill only
      for(int v : nb) deg[v] = wdeg(v);
      adj[best].clear();
    }
    for(int i = 0; i < n; i++) loc[nodes[i]] = -1;
  }

  void run(const std::vector<int>& all_nodes)
  {
    inset.assign(G.n, 0); vis.assign(G.n, 0);
    std::vector<int> dist(G.n, 0);
    int setctr = 0;
    // explicit stack of node sets; a frame with `sep` set flushes a separator
    struct Frame { std::vector<int> nodes; bool is_sep; };
    std::vector<Frame> stack;
    stack.push_back({all_nodes, false});
    std::vector<int> order, lvl_start, comp;
    while(!stack.empty())
    {
      Frame f = std::move(stack.back());
      stack.pop_back();
      if(f.is_sep) { out.insert(out.end(), f.nodes.begin(), f.nodes.end()); continue; }
      if(f.nodes.empty()) continue;
      // stamp the set
      const int setstamp = ++setctr;
      for(int u : f.nodes) inset[u] = setstamp;
      // split into connected components; process each
      std::vector<std::vector<int>> comps;
      ++stamp;
      const int cstamp = stamp;
      for(int u : f.nodes)
      {
        if(vis[u] == cstamp) continue;
        comp.clear(); comp.push_back(u); vis[u] = cstamp;
        for(size_t h = 0; h < comp.size(); h++)
        {
          const int a = comp[h];
          for(int e = G.ptr[a]; e < G.ptr[a+1]; e++)
          {
            const int v = G.adj[e];
            if(inset[v] == setstamp && vis[v] != cstamp) { vis[v] = cstamp; comp.push_back(v); }
          }
        }
        comps.push_back(comp);
      }
      if(comps.size() > 1)
      {
        // push in reverse so that the first component is ordered first
        for(size_t c = comps.size(); c-- > 0;) stack.push_back({std::move(comps[c]), false});
        continue;
      }
      std::vector<int>& nodes = comps[0];
      long wt = 0; for(int u : nodes) wt += G.w[u];
      if(wt <= leaf_w || nodes.size() <= 3) { std::sort(nodes.begin(), nodes.end()); order_leaf(nodes); continue; }
      // pseudo-peripheral root: a few BFS sweeps from the min-degree node
      int root = nodes[0];
      for(int u : nodes) if(G.wdeg[u] < G.wdeg[root] || (G.wdeg[u] == G.wdeg[root] && u < root)) root = u;
      int ecc = bfs(root, setstamp, order, lvl_start, dist);
      for(int sweep = 0; sweep < 3; sweep++)
      {
        // candidate: min-degree node of the last level
        const int nl = (int)lvl_start.size() - 1;
        int cand = order[lvl_start[nl-1]];
        for(int i = lvl_start[nl-1]; i < lvl_start[nl]; i++)
          if(G.wdeg[order[i]] < G.wdeg[cand]) cand = order[i];
        std::vector<int> o2, l2;
        const int e2 = bfs(cand, setstamp, o2, l2, dist);
        if(e2 > ecc) { ecc = e2; root = cand; order.swap(o2); lvl_start.swap(l2); }
        else { bfs(root, setstamp, order, lvl_start, dist); break; }
      }
      const int nl = (int)lvl_start.size() - 1;
      if(nl < 3) { std::sort(nodes.begin(), nodes.end()); order_leaf(nodes); continue; }
      // level weights, choose the lightest level whose prefix weight is within [0.3,0.7]
      std::vector<long> lw(nl, 0);
      for(int l = 0; l < nl; l++) for(int i = lvl_start[l]; i < lvl_start[l+1]; i++) lw[l] += G.w[order[i]];
      long pre = 0; int best = -1; double bestscore = 0;
      for(int l = 0; l < nl; l++)
      {
        const double before = (double)pre/(double)wt, after = (double)(wt - pre - lw[l])/(double)wt;
        pre += lw[l];
        if(l == 0 || l == nl-1) continue;
        if(before < 0.25 || after < 0.25) continue;
        // score: small separators first, balance as tie-break
        const double score = (double)lw[l]*(1.0 + 0.5*fabs(before - after));
        if(best < 0 || score < bestscore) { best = l; bestscore = score; }
      }
      if(best < 0)
      {
        // no balanced level: take the middle one by weight
        pre = 0;
        for(int l = 0; l < nl; l++) { pre += lw[l]; if(2*pre >= wt) { best = l; break; } }
        if(best <= 0) best = 1;
        if(best >= nl-1) best = nl-2;
      }
      std::vector<int> A(order.begin(), order.begin() + lvl_start[best]);
      std::vector<int> Sp(order.begin() + lvl_start[best], order.begin() + lvl_start[best+1]);
      std::vector<int> B(order.begin() + lvl_start[best+1], order.end());
      if(A.empty() || B.empty()) { std::sort(nodes.begin(), nodes.end()); order_leaf(nodes); continue; }
      std::sort(Sp.begin(), Sp.end());
      // order: A ..., B ..., then the separator (stack is LIFO)
      stack.push_back({std::move(Sp), true});
      stack.push_back({std::move(B), false});
      stack.push_back({std::move(A), false});
    }
  }
};

} // namespace

#define SYM_FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); return 1; } while(0)

int sym_analyze(SymHost& S, int N, int M, const int* cp, const int* ri, int row0, int row1,
                char* err, int errlen, int part_rank, int part_nranks)
{
  const bool partition = part_nranks > 1;
  if(partition) { row0 = 0; row1 = M; }           // the rows of a rank are chosen below, not given
  // DOGLEG_AMD_SYM_DEBUG >= 2: wall time of every step on stderr
  const bool sym_time = env_int("DOGLEG_AMD_SYM_DEBUG", 0) >= 2;
  auto sym_t0 = std::chrono::steady_clock::now();
  const char* sym_what = "setup";
  bool sym_parallel = false;                 // inside the threaded section: the step clock is not kept
  auto sym_tick = [&](const char* next) {
    if(sym_parallel) return;
    if(sym_time)
    {
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "sym_analyze: %-40s %8.1f ms\n", sym_what, std::chrono::duration<double, std::milli>(t1 - sym_t0).count());
      sym_t0 = t1;
    }
    sym_what = next; };
#define SYM_TICK(name) sym_tick(name)

  S = SymHost();
  S.N = N; S.M = M; S.nnz = cp[M]; S.row0 = row0; S.row1 = row1;
  S.part_rank = partition ? part_rank : 0; S.part_nranks = partition ? part_nranks : 1;
  if(cp[0] != 0) SYM_FAIL("Jt column pointers must start at 0");

  SYM_TICK("1 var-blocks");
  // ---------------------------------------------------------- 1. var-blocks
  std::vector<char> cut(N + 1, 0);
  cut[0] = cut[N] = 1;
  for(int r = 0; r < M; r++)
  {
    const int a = cp[r], b = cp[r+1];
    if(b < a) SYM_FAIL("Jt column pointers must be non-decreasing (row %d)", r);
    if(b - a > 65535) SYM_FAIL("measurement row %d has %d non-zeros; at most 65535 are supported", r, b - a);
    for(int q = a; q < b; q++)
    {
      const int i = ri[q];
      if(i < 0 || i >= N) SYM_FAIL("row index %d out of range in measurement %d", i, r);
      if(q > a && ri[q-1] >= i) SYM_FAIL("row indices of measurement %d are not strictly ascending", r);
      if(q == a   || ri[q-1] != i-1) cut[i] = 1;
      if(q == b-1 || ri[q+1] != i+1) cut[i+1] = 1;
    }
  }
  std::vector<int> vb_of(N);
  {
    int start = 0;
    for(int j = 1; j <= N; j++)
      if(cut[j] || j - start == VB_MAX) { cut[j] = 1; start = j; }
    S.vb_start.clear();
    for(int j = 0; j <= N; j++) if(cut[j]) S.vb_start.push_back(j);
    S.nvb = (int)S.vb_start.size() - 1;
    for(int v = 0; v < S.nvb; v++) for(int j = S.vb_start[v]; j < S.vb_start[v+1]; j++) vb_of[j] = v;
  }
  const int nvb = S.nvb;
  auto vbw = [&](int v) { return S.vb_start[v+1] - S.vb_start[v]; };

  SYM_TICK("2 row-blocks");
  // ---------------------------------------------------------- 2. row-blocks
  std::vector<RowBlock> rbs;
  std::vector<int> rb_vb, rb_off;
  {
    int cur = -1;
    for(int r = 0; r < M; r++)
    {
      const int len = cp[r+1] - cp[r];
      if(len == 0) { cur = -1; continue; }
      bool fresh = (cur < 0) || r == row0 || r == row1;
      if(!fresh)
      {
        const RowBlock& b = rbs[cur];
        if(b.len != len || b.nrows == RB_MAX || b.r0 + b.nrows != r ||
           memcmp(&ri[b.base], &ri[cp[r]], sizeof(int)*(size_t)len) != 0) fresh = true;
      }
      if(fresh)
      {
        RowBlock b; b.r0 = r; b.nrows = 1; b.len = len; b.base = cp[r]; b.vptr = (int)rb_vb.size();
        b.local = (r >= row0 && r < row1);
        b.lbase = b.base - cp[row0]; b.lr0 = r - row0;
        int e = 0;
        while(e < len)
        {
          const int v = vb_of[ri[b.base + e]];
          rb_vb.push_back(v); rb_off.push_back(e);
          e += S.vb_start[v+1] - ri[b.base + e];     // remainder of the block from this entry
        }
        b.nvb = (int)rb_vb.size() - b.vptr;
        rbs.push_back(b);
        cur = (int)rbs.size() - 1;
      }
      else rbs[cur].nrows++;
    }
  }
  const int nrb = (int)rbs.size();

  SYM_TICK("3 block graph of JtJ");
  // ------------------------------------------------ 3. block graph of JtJ
  Graph G; G.n = nvb; G.w.resize(nvb);
  for(int v = 0; v < nvb; v++) G.w[v] = vbw(v);
  {
    // inverted index vb -> row-blocks; skip row-blocks with a pattern identical to the previous one
    std::vector<int> cnt(nvb + 1, 0);
    std::vector<char> dup(nrb, 0);
    for(int b = 1; b < nrb; b++)
      if(rbs[b].nvb == rbs[b-1].nvb &&
         memcmp(&rb_vb[rbs[b].vptr], &rb_vb[rbs[b-1].vptr], sizeof(int)*(size_t)rbs[b].nvb) == 0) dup[b] = 1;
    for(int b = 0; b < nrb; b++) if(!dup[b]) for(int k = 0; k < rbs[b].nvb; k++) cnt[rb_vb[rbs[b].vptr + k] + 1]++;
    for(int v = 0; v < nvb; v++) cnt[v+1] += cnt[v];
    std::vector<int> inv(cnt[nvb]), nxt(cnt.begin(), cnt.end() - 1);
    for(int b = 0; b < nrb; b++) if(!dup[b]) for(int k = 0; k < rbs[b].nvb; k++) inv[nxt[rb_vb[rbs[b].vptr + k]]++] = b;
    std::vector<int> mark(nvb, -1);
    G.ptr.assign(nvb + 1, 0);
    std::vector<int> tmp;
    for(int v = 0; v < nvb; v++)
    {
      tmp.clear();
      mark[v] = v;
      for(int a = cnt[v]; a < cnt[v+1]; a++)
      {
        const RowBlock& b = rbs[inv[a]];
        for(int k = 0; k < b.nvb; k++)
        {
          const int u = rb_vb[b.vptr + k];
          if(mark[u] != v) { mark[u] = v; tmp.push_back(u); }
        }
      }
      std::sort(tmp.begin(), tmp.end());
      G.adj.insert(G.adj.end(), tmp.begin(), tmp.end());
      G.ptr[v+1] = (int)G.adj.size();
    }
    G.wdeg.assign(nvb, 0);
    for(int v = 0; v < nvb; v++) for(int e = G.ptr[v]; e < G.ptr[v+1]; e++) G.wdeg[v] += G.w[G.adj[e]];
  }
  // nnz(tril JtJ)
  S.nnz_JtJ_lower = 0;
  for(int v = 0; v < nvb; v++)
  {
    const long w = G.w[v];
    S.nnz_JtJ_lower += w*(w+1)/2;
    for(int e = G.ptr[v]; e < G.ptr[v+1]; e++) if(G.adj[e] > v) S.nnz_JtJ_lower += w*G.w[G.adj[e]];
  }

  SYM_TICK("4 ordering");
  // ------------------------------------------------------------ 4. ordering
  std::vector<int> border;            // position -> vb
  {
    const double dense_thr = std::max(16.0, 10.0*sqrt((double)N));
    std::vector<int> sparse_nodes, dense_nodes;
    Graph H;                           // graph without dense nodes
    std::vector<char> is_dense(nvb, 0);
    for(int v = 0; v < nvb; v++) if((double)G.wdeg[v] > dense_thr && nvb > 8) is_dense[v] = 1;
    for(int v = 0; v < nvb; v++) (is_dense[v] ? dense_nodes : sparse_nodes).push_back(v);
    H.n = nvb; H.w = G.w; H.ptr.assign(nvb + 1, 0);
    for(int v = 0; v < nvb; v++)
    {
      if(!is_dense[v]) for(int e = G.ptr[v]; e < G.ptr[v+1]; e++) if(!is_dense[G.adj[e]]) H.adj.push_back(G.adj[e]);
      H.ptr[v+1] = (int)H.adj.size();
    }
    H.wdeg = G.wdeg;
    Orderer O(H);
    O.run(sparse_nodes);
    border = O.out;
    // dense nodes last, lightest first
    std::sort(dense_nodes.begin(), dense_nodes.end(), [&](int a, int b) {
      if(G.wdeg[a] != G.wdeg[b]) return G.wdeg[a] < G.wdeg[b];
      return a < b; });
    border.insert(border.end(), dense_nodes.begin(), dense_nodes.end());
    if((int)border.size() != nvb) SYM_FAIL("internal error: ordering lost blocks (%zu of %d)", border.size(), nvb);
  }
  std::vector<int> bpos(nvb);
  for(int k = 0; k < nvb; k++) bpos[border[k]] = k;

  SYM_TICK("5 block-level symbolic factorisation");
  // --------------------------------- 5. block-level symbolic factorisation
  std::vector<std::vector<int>> st(nvb);   // struct of block column (positions > j), sorted
  std::vector<int> parent(nvb, -1);
  std::vector<int> nchild(nvb, 0);
  auto symbolic = [&]() {
    std::vector<int> head(nvb, -1), next(nvb, -1), mark(nvb, -1);
    std::vector<int> tmp;
    std::fill(parent.begin(), parent.end(), -1);
    std::fill(nchild.begin(), nchild.end(), 0);
    for(int j = 0; j < nvb; j++)
    {
      tmp.clear();
      mark[j] = j;
      const int v = border[j];
      for(int e = G.ptr[v]; e < G.ptr[v+1]; e++)
      {
        const int q = bpos[G.adj[e]];
        if(q > j && mark[q] != j) { mark[q] = j; tmp.push_back(q); }
      }
      for(int c = head[j]; c >= 0; c = next[c])
      {
        for(int q : st[c]) if(q > j && mark[q] != j) { mark[q] = j; tmp.push_back(q); }
      }
      std::sort(tmp.begin(), tmp.end());
      st[j] = tmp;
      if(!tmp.empty()) { parent[j] = tmp[0]; next[j] = head[tmp[0]]; head[tmp[0]] = j; nchild[tmp[0]]++; }
    }
  };
  symbolic();
  // Leaves of the elimination tree with IDENTICAL structure (e.g. the points seen by
  // the same set of cameras) are independent of each other; placed next to each other
  // they form one supernode with a block-diagonal top -> far fewer, wider panels
  // (rank-k updates instead of k rank-3 ones).  Moving a leaf earlier is always a
  // valid elimination order.
  if(true)
  {
    std::vector<int> leaves;
    for(int j = 0; j < nvb; j++) if(nchild[j] == 0 && !st[j].empty()) leaves.push_back(j);
    std::sort(leaves.begin(), leaves.end(), [&](int x, int y) {
      if(st[x] != st[y]) return st[x] < st[y];
      return x < y; });
    std::vector<int> group_of(nvb, -1), group_first;
    std::vector<std::vector<int>> members;
    for(size_t i = 0; i < leaves.size();)
    {
      size_t k = i + 1;
      while(k < leaves.size() && st[leaves[k]] == st[leaves[i]]) k++;
      if(k - i > 1)
      {
        std::vector<int> m(leaves.begin() + i, leaves.begin() + k);      // ascending positions
        for(int x : m) group_of[x] = (int)members.size();
        members.push_back(m);
      }
      i = k;
    }
    // New order: every leaf first (grouped by structure, groups and single leaves by
    // first position), then the remaining nodes in a POSTORDER of the elimination tree
    // restricted to them.  With the leaves out of the way a chain of the tree (e.g. the
    // cameras of one dissection leaf, whose other children are all points) becomes a run
    // of consecutive columns and merges into one supernode.
    {
      std::vector<char> is_leaf(nvb, 0);
      for(int j : leaves) is_leaf[j] = 1;
      std::vector<int> nb; nb.reserve(nvb);
      std::vector<char> done(members.size(), 0);
      for(int j = 0; j < nvb; j++)
      {
        if(!is_leaf[j]) continue;
        const int g = group_of[j];
        if(g < 0) { nb.push_back(border[j]); continue; }
        if(done[g]) continue;
        done[g] = 1;
        for(int x : members[g]) nb.push_back(border[x]);
      }
      // children lists of the non-leaf forest
      std::vector<int> chead(nvb, -1), cnext(nvb, -1);
      std::vector<int> roots;
      for(int j = nvb - 1; j >= 0; j--)
      {
        if(is_leaf[j]) continue;
        if(parent[j] >= 0) { cnext[j] = chead[parent[j]]; chead[parent[j]] = j; }   // ascending child lists
        else roots.push_back(j);
      }
      std::sort(roots.begin(), roots.end());
      std::vector<int> stack, it(nvb, -2);
      for(int r : roots)
      {
        stack.push_back(r);
        while(!stack.empty())
        {
          const int u = stack.back();
          if(it[u] == -2) it[u] = chead[u];
          if(it[u] >= 0) { const int c = it[u]; it[u] = cnext[c]; stack.push_back(c); }
          else { nb.push_back(border[u]); stack.pop_back(); }
        }
      }
      if((int)nb.size() != nvb) SYM_FAIL("internal error: postorder lost blocks (%zu of %d)", nb.size(), nvb);
      border.swap(nb);
      for(int k = 0; k < nvb; k++) bpos[border[k]] = k;
      symbolic();
    }
  }
  std::vector<long> stw(nvb, 0);          // scalar weight of struct
  for(int j = 0; j < nvb; j++) for(int q : st[j]) stw[j] += G.w[border[q]];
  S.nnz_L = 0; S.factor_flops = 0;
  for(int j = 0; j < nvb; j++)
  {
    const long w = G.w[border[j]];
    S.nnz_L += w*(w+1)/2 + w*stw[j];
    for(long a = 0; a < w; a++) { const double c = (double)(w - a + stw[j]); S.factor_flops += c*c; }
  }

  SYM_TICK("6 supernodes");
  // ------------------------------------------------------- 6. supernodes
  std::vector<int> sn_b0;                 // first block position of each supernode (+ sentinel)
  std::vector<char> sn_bd;                // supernode made of sibling leaves only: block-diagonal top
  {
    const int relax = 25;
    const int sib_w = 64;
    const int split_w = 32;   // columns from which a run stays a supernode of its own beside its parent's other children
    const long chain_cap = PANEL_CAP;   // width cap of chain supernodes: W*(W+64)
    const bool lds_split = env_int("DOGLEG_AMD_LDS_SPLIT", 1) != 0;     // (0: row-sliced separators outside the multifrontal region, rounds 1 - 3: a path of its own, kept under test on config #5)
    const long lds_split_minw = 32;
    // width of the fundamental supernode (maximal chain of exactly nested block columns) starting at j:
    // a relaxed merge across a structure change takes that whole run or nothing -- stopping in the
    // middle of it at the width cap leaves fragments that cost an elimination-tree level each
    std::vector<long> run_w(nvb + 1, 0);
    for(int j = nvb - 1; j >= 0; j--)
    {
      const bool exact_next = j + 1 < nvb && parent[j] == j + 1 && st[j].size() == st[j+1].size() + 1;
      run_w[j] = G.w[border[j]] + (exact_next ? run_w[j+1] : 0);
    }
    // children lists and the longest chain of columns below (and including) every block column
    std::vector<int> kid_head(nvb, -1), kid_next(nvb, -1);
    std::vector<long> crit(nvb, 0);
    for(int j = 0; j < nvb; j++)
    {
      crit[j] += G.w[border[j]];
      if(parent[j] >= 0)
      {
        kid_next[j] = kid_head[parent[j]]; kid_head[parent[j]] = j;
        crit[parent[j]] = std::max(crit[parent[j]], crit[j]);
      }
    }
    int a = 0;
    long W = G.w[border[0]];              // current width
    long true_nnz = W*stw[0];             // sum_j w_j * |struct_j| (scalar) for columns in the supernode
    long below_own = 0;                   // helper: sum_j w_j * (own cols after j)
    sn_b0.push_back(0);
    bool sib_only = true; int nmerge = 0;
    for(int j = 0; j + 1 <= nvb; j++)
    {
      bool merge = false;
      if(j + 1 < nvb && parent[j] == j + 1)
      {
        const long w1 = G.w[border[j+1]];
        const long Wn = W + w1, Rn = stw[j+1];
        // zeros in the rectangular-below + trapezoid-own storage if merged
        const long tn = true_nnz + w1*stw[j+1];
        // stored (excluding the dense diagonal block): every column holds Rn below rows + own cols after it
        long own_after = below_own + W*w1;    // each existing column gains w1 own rows after it
        const long stored = Wn*Rn + own_after;
        const long zeros = stored - tn;
        const bool exact = (st[j].size() == st[j+1].size() + 1);
        // a panel is factored in row slices (>= 64 rows each) that all carry the w x w top block
        // (... unless it has fewer rows below than that: the root of the tree, whose panel is its top block)
        const long below = std::min<long>(64, Rn);
        const bool fits = (Wn*(Wn + below) <= chain_cap) && Wn <= SN_WMAX;
        const long Wrun = W + run_w[j+1];
        const bool run_fits = (Wrun*(Wrun + below) <= chain_cap) && Wrun <= SN_WMAX;
        // A wide run of columns is NOT merged into a parent that has other children: merged, it is eliminated
        // after all of them (the merged supernode waits for every child); on its own it is eliminated beside
        // them.  The top of a nested dissection: the separator of one half and the root separator -- 60
        // columns less on the critical path of config #4.
        // Costs in columns on the critical path (crit[c]: the longest chain of columns ending with block c):
        // merged, the supernode starts when ALL children are done -- max(crit[j], sib + W) to its end --, apart
        // the run ends at crit[j] and the parent starts at max(crit[j], sib) plus one more hand-off (~split_w
        // columns' worth).  Siblings that are leaves (points) never make it worth it.
        long sib = 0;
        if(nchild[j+1] >= 2)
          for(int c = kid_head[j+1]; c >= 0; c = kid_next[c]) if(c != j) sib = std::max(sib, crit[c]);
        const bool beside_siblings = nchild[j+1] >= 2 && std::min<long>(W, sib + W - crit[j]) > split_w;
        // A chain whose panel would not fit LDS whole is cut where it still does, once it is wide enough to be worth a
        // workgroup (config #5: separators of 96 columns with 194 rows below, 222 KB -- as one supernode they are cut into
        // ROW slices, which keeps every level above the leaves out of the multifrontal one-launch region; as 64 + 32
        // columns both panels fit and the whole top of the tree is one launch).
        const long pan_n = ((Wn + Rn + 1 + 1) & ~1L)*Wn;
        const bool lds_cut = lds_split && W >= lds_split_minw && pan_n > slice_cap() && Rn + 1 <= 255;
        if(fits && !beside_siblings && !lds_cut && (exact || (run_fits && (Wn <= 16 || zeros*100 <= (long)relax*(stored + Wn*Wn)))))
        {
          merge = true; sib_only = false; nmerge++;
          W = Wn; true_nnz = tn; below_own = own_after;
        }
      }
      if(!merge && j + 1 < nvb && nchild[j] == 0 && nchild[j+1] == 0 && !st[j].empty() && st[j] == st[j+1])
      {
        // sibling leaves with identical structure: block-diagonal top, same rows below
        const long w1 = G.w[border[j+1]];
        const long Wn = W + w1, Rn = stw[j+1];
        if(Wn <= sib_w && (Wn + Rn + 1)*Wn <= PANEL_CAP)
        {
          merge = true; nmerge++;
          true_nnz += w1*stw[j+1]; below_own += W*w1; W = Wn;
        }
      }
      if(!merge && j + 1 < nvb)
      {
        sn_bd.push_back((sib_only && nmerge > 0) ? 1 : 0); sib_only = true; nmerge = 0;
        a = j + 1; sn_b0.push_back(a);
        W = G.w[border[a]]; true_nnz = W*stw[a]; below_own = 0;
      }
    }
    sn_bd.push_back((sib_only && nmerge > 0) ? 1 : 0);
    sn_b0.push_back(nvb);
  }
  S.nsn = (int)sn_b0.size() - 1;
  const int nsn = S.nsn;

  SYM_TICK("7 scalar-level layout");
  // ------------------------------------------- 7. scalar-level layout
  std::vector<int> colstart(nvb + 1, 0);  // scalar position of block position
  for(int k = 0; k < nvb; k++) colstart[k+1] = colstart[k] + G.w[border[k]];
  S.perm.resize(N); S.iperm.resize(N);
  for(int k = 0; k < nvb; k++)
    for(int a = 0; a < G.w[border[k]]; a++)
    { S.perm[colstart[k] + a] = S.vb_start[border[k]] + a; S.iperm[S.vb_start[border[k]] + a] = colstart[k] + a; }
  std::vector<int> sn_of_b(nvb);
  S.sn_c0.resize(nsn + 1); S.sn_rowptr.assign(nsn + 1, 0); S.sn_lx.assign(nsn + 1, 0); S.sn_scr.assign(nsn + 1, 0);
  // block-level below structure of a supernode = struct of its last block column
  auto sn_last = [&](int s) { return sn_b0[s+1] - 1; };
  for(int s = 0; s < nsn; s++)
  {
    for(int k = sn_b0[s]; k < sn_b0[s+1]; k++) sn_of_b[k] = s;
    S.sn_c0[s] = colstart[sn_b0[s]];
  }
  S.sn_c0[nsn] = N;
  // members (var-blocks) of the block-diagonal supernodes
  S.sn_bd_ptr.assign(nsn + 1, 0);
  for(int s2 = 0; s2 < nsn; s2++)
  {
    if(sn_bd[s2]) for(int k = sn_b0[s2]; k < sn_b0[s2+1]; k++) S.sn_bd_col.push_back(colstart[k] - colstart[sn_b0[s2]]);
    S.sn_bd_ptr[s2+1] = (int)S.sn_bd_col.size();
  }
  S.max_panel = 0;
  for(int s = 0; s < nsn; s++)
  {
    const int w = S.sn_c0[s+1] - S.sn_c0[s];
    const long r = stw[sn_last(s)];
    const long nrows = w + r + 1;          // + the augmented right-hand-side row (see sparse_symbolic.h)
    S.sn_rowptr[s+1] = S.sn_rowptr[s] + (int)nrows;
    S.sn_lx[s+1] = S.sn_lx[s] + nrows*(long)w;
    S.sn_scr[s+1] = S.sn_scr[s] + (int)r;
    if(nrows*w > S.max_panel) S.max_panel = (int)(nrows*w);
  }
  S.lx_size = S.sn_lx[nsn]; S.scr_size = S.sn_scr[nsn];
  S.sn_rows.resize(S.sn_rowptr[nsn]);
  // per supernode: block-level row offsets of the below blocks (for lookups)
  std::vector<int> belowoff_ptr(nsn + 1, 0);
  for(int s = 0; s < nsn; s++) belowoff_ptr[s+1] = belowoff_ptr[s] + (int)st[sn_last(s)].size();
  std::vector<int> belowoff(belowoff_ptr[nsn]);
  for(int s = 0; s < nsn; s++)
  {
    int* rows = &S.sn_rows[S.sn_rowptr[s]];
    const int w = S.sn_c0[s+1] - S.sn_c0[s];
    int k = 0;
    for(; k < w; k++) rows[k] = S.sn_c0[s] + k;
    int bi = belowoff_ptr[s];
    for(int q : st[sn_last(s)])
    {
      belowoff[bi++] = k;
      for(int a = 0; a < G.w[border[q]]; a++) rows[k++] = colstart[q] + a;
    }
    rows[k++] = N;                         // augmented row: virtual variable N, present in every panel
  }
  S.diagpos.resize(N); S.col_sn.resize(N);
  for(int s = 0; s < nsn; s++)
  {
    const int w = S.sn_c0[s+1] - S.sn_c0[s];
    const long ld = S.sn_rowptr[s+1] - S.sn_rowptr[s];
    for(int c = 0; c < w; c++) { S.diagpos[S.sn_c0[s] + c] = S.sn_lx[s] + c + c*ld; S.col_sn[S.sn_c0[s] + c] = s; }
  }
  // row offset of block position q inside supernode t's row list (-1 if absent)
  auto rowoff_in = [&](int t, int q) -> int {
    if(q >= sn_b0[t] && q < sn_b0[t+1]) return colstart[q] - S.sn_c0[t];
    const std::vector<int>& bl = st[sn_last(t)];
    auto it = std::lower_bound(bl.begin(), bl.end(), q);
    if(it == bl.end() || *it != q) return -1;
    return belowoff[belowoff_ptr[t] + (int)(it - bl.begin())];
  };

  // levels
  S.sn_level.assign(nsn, 0);
  for(int s = 0; s < nsn; s++)
  {
    const std::vector<int>& bl = st[sn_last(s)];
    if(bl.empty()) continue;
    const int p = sn_of_b[bl[0]];
    if(S.sn_level[p] < S.sn_level[s] + 1) S.sn_level[p] = S.sn_level[s] + 1;
  }
  S.nlevels = 0;
  for(int s = 0; s < nsn; s++) S.nlevels = std::max(S.nlevels, S.sn_level[s] + 1);
  S.lvl_ptr.assign(S.nlevels + 1, 0);
  for(int s = 0; s < nsn; s++) S.lvl_ptr[S.sn_level[s] + 1]++;
  for(int l = 0; l < S.nlevels; l++) S.lvl_ptr[l+1] += S.lvl_ptr[l];
  S.lvl_sn.resize(nsn);
  {
    std::vector<int> nx(S.lvl_ptr.begin(), S.lvl_ptr.end() - 1);
    for(int s = 0; s < nsn; s++) S.lvl_sn[nx[S.sn_level[s]]++] = s;
  }

  // ---- subtree partition (multi-GPU, SURVEY 8e "point ownership"): the elimination tree is cut
  // above level `cut_level`; every subtree below the cut belongs to ONE rank, which holds the
  // measurement rows whose first-eliminated variable lies in it, assembles, factors and solves it
  // alone -- a row's variables are a clique of JtJ, so they all sit on one root path and a subtree's
  // panels receive contributions from its owner's rows only.  The supernodes above the cut are
  // replicated: their panels (assembled entries + the owners' updates, both partial sums) and the
  // update matrices handed up across the cut are summed over the ranks once per factorisation.
  std::vector<int> sn_parent0(nsn, -1);
  for(int d = 0; d < nsn; d++) { const std::vector<int>& bl = st[sn_last(d)]; if(!bl.empty()) sn_parent0[d] = sn_of_b[bl[0]]; }
  S.sn_owner.assign(nsn, -1);
  S.cut_level = -1;
  if(partition)
  {
    const int oversub = 1;
    for(int l = S.nlevels - 1; l >= 0; l--)
      if(S.lvl_ptr[l+1] - S.lvl_ptr[l] >= part_nranks*oversub) { S.cut_level = l; break; }
    const int Lc = S.cut_level;
    // weight of a subtree: panel entries (a proxy for its assembly, factorisation and solve work)
    std::vector<double> wt(nsn, 0.0);
    for(int d = 0; d < nsn; d++)
    {
      wt[d] += (double)(S.sn_rowptr[d+1] - S.sn_rowptr[d])*(S.sn_c0[d+1] - S.sn_c0[d]);
      if(sn_parent0[d] >= 0 && S.sn_level[sn_parent0[d]] <= Lc) wt[sn_parent0[d]] += wt[d];     // parents come later
    }
    std::vector<int> roots;
    for(int d = 0; d < nsn; d++)
      if(S.sn_level[d] <= Lc && (sn_parent0[d] < 0 || S.sn_level[sn_parent0[d]] > Lc)) roots.push_back(d);
    std::sort(roots.begin(), roots.end(), [&](int a, int b) { if(wt[a] != wt[b]) return wt[a] > wt[b]; return a < b; });
    std::vector<double> load(part_nranks, 0.0);
    for(int d : roots)
    {
      int best = 0;
      for(int r = 1; r < part_nranks; r++) if(load[r] < load[best]) best = r;
      S.sn_owner[d] = best; load[best] += wt[d];
    }
    for(int d = nsn - 1; d >= 0; d--)
      if(S.sn_level[d] <= Lc && S.sn_owner[d] < 0) S.sn_owner[d] = S.sn_owner[sn_parent0[d]];
    // rows: the row-blocks whose first-eliminated variable lies in one of this rank's subtrees; rows
    // that only touch replicated variables are dealt out in turn
    int lb = 0, lr = 0, turn = 0;
    S.part_rows.clear();
    for(int bi = 0; bi < nrb; bi++)
    {
      RowBlock& b = rbs[bi];
      int qmin = nvb;
      for(int x = 0; x < b.nvb; x++) qmin = std::min(qmin, bpos[rb_vb[b.vptr + x]]);
      int own = S.sn_owner[sn_of_b[qmin]];
      if(own < 0) own = (turn++) % part_nranks;
      b.local = own == part_rank;
      b.lbase = lb; b.lr0 = lr;
      if(b.local) { lb += b.nrows*b.len; lr += b.nrows; for(int r = 0; r < b.nrows; r++) S.part_rows.push_back(b.r0 + r); }
    }
  }
  auto mine = [&](int s) { return S.sn_owner[s] < 0 || S.sn_owner[s] == part_rank; };
  // the supernodes this rank works on, by level (all of them on a single rank)
  S.xl_ptr.assign(S.nlevels + 1, 0);
  S.xl_sn.clear();
  for(int l = 0; l < S.nlevels; l++)
  {
    for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1]; i++) if(mine(S.lvl_sn[i])) S.xl_sn.push_back(S.lvl_sn[i]);
    S.xl_ptr[l+1] = (int)S.xl_sn.size();
  }

  // factor work list: (supernode, slice of its below rows).  Panels larger than the LDS
  // budget are cut into row slices; every slice workgroup also holds the w x w top block
  // and factors it redundantly, only slice 0 publishes it (into top_scr, copied back at
  // the end: no other kernel of the factorisation reads a top block).
  {
    S.fw_lvl_ptr.assign(S.nlevels + 1, 0);
    S.sn_top.assign(nsn, -1);
    S.top_size = 0;
    for(int l = 0; l < S.nlevels; l++)
    {
      for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1]; i++)
      {
        const int s = S.lvl_sn[i];
        const int w = S.sn_c0[s+1] - S.sn_c0[s];
        const int nrows = S.sn_rowptr[s+1] - S.sn_rowptr[s];
        const int below = nrows - w;
        int nsl = 1;
        if((long)((nrows + 1) & ~1)*w > slice_cap())
        {
          int rpw = slice_cap()/w - w - 1; if(rpw < 1) rpw = 1;
          nsl = (below + rpw - 1)/rpw;
        }
        // (the slicing of every supernode is recorded -- it decides where the multifrontal region
        // starts, identically on every rank --, work items only for this rank's supernodes)
        if(nsl > 1) { S.sn_top[s] = S.top_size; S.top_size += (int64_t)w*w; if(mine(s)) S.ms_sn.push_back(s); }
        if(!mine(s)) continue;
        const int per = (below + nsl - 1)/nsl;
        for(int k = 0; k < nsl; k++)
        {
          S.fw_sn.push_back(s); S.fw_r0.push_back(std::min(below, k*per)); S.fw_r1.push_back(std::min(below, (k+1)*per));
        }
        S.fw_lvl_ptr[l+1] += nsl;
      }
    }
    for(int l = 0; l < S.nlevels; l++) S.fw_lvl_ptr[l+1] += S.fw_lvl_ptr[l];
  }

  if(env_int("DOGLEG_AMD_SYM_DEBUG", 0))
  {
    for(int l = 0; l < S.nlevels; l++)
    {
      long n = 0, wsum = 0, rsum = 0, nofit = 0; int wmin = 1 << 30, wmax = 0, rmax = 0;
      for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1]; i++)
      {
        const int s = S.lvl_sn[i];
        const int w = S.sn_c0[s+1] - S.sn_c0[s], nr = S.sn_rowptr[s+1] - S.sn_rowptr[s];
        n++; wsum += w; rsum += nr; wmin = std::min(wmin, w); wmax = std::max(wmax, w); rmax = std::max(rmax, nr);
        const long mb = nr - w;
        (void)mb;
        if(sym_w_split(w, nr) < 0) nofit++;
      }
      fprintf(stderr, "level %2d: %5ld supernodes  w min/avg/max %d/%.1f/%d  nrows avg/max %.1f/%d  slices %d  update matrix not in LDS: %ld\n", l, n, wmin,
              (double)wsum/n, wmax, (double)rsum/n, rmax, S.fw_lvl_ptr[l+1] - S.fw_lvl_ptr[l], nofit);
    }
  }

  // (steps 8, 9a, 9b and 10 only read what the steps before them built and write disjoint parts of S:
  // they run on four threads, see below)
  auto step8 = [&](char* err, int errlen) -> int {
  // ------------------------------------ 8. factor update schedule
  {
    struct Sub { int lvl, t, q, d, ka, rel; };
    std::vector<Sub> subs;
    // multifrontal region: the levels from mf_level0 up, provided none of their supernodes is
    // cut into slices or has more than MF_MAXM rows below its diagonal block (one workgroup
    // adds a child's whole update matrix).  DOGLEG_AMD_MF_LEVEL: lowest level allowed in the
    // region, < 0 turns it off.
    S.mf_level0 = S.nlevels;
    {
      const int mf_req = env_int("DOGLEG_AMD_MF_LEVEL", 1), mf_maxm = 255;
      for(int l = S.nlevels - 1; mf_req >= 0 && l >= mf_req; l--)
      {
        bool ok = true;
        for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1] && ok; i++)
        {
          const int d = S.lvl_sn[i];
          const int mb = S.sn_rowptr[d+1] - S.sn_rowptr[d] - (S.sn_c0[d+1] - S.sn_c0[d]);
          if(S.sn_top[d] >= 0 || mb > mf_maxm) ok = false;
        }
        if(!ok) break;
        S.mf_level0 = l;
      }
    }
    S.sn_prel.assign(nsn, -1);
    std::vector<int> sn_parent(nsn, -1);
    for(int d = 0; d < nsn; d++)
    {
      const std::vector<int>& bl = st[sn_last(d)];
      if(bl.empty()) continue;
      const bool mf_src = S.sn_level[d] >= S.mf_level0;
      const int wd = S.sn_c0[d+1] - S.sn_c0[d];
      const int nrows_d = S.sn_rowptr[d+1] - S.sn_rowptr[d];
      size_t i = 0;
      int krow = wd;                       // row index in d of block bl[i]
      while(i < bl.size())
      {
        const int t = sn_of_b[bl[i]];
        const int k0 = krow;
        const int relbase = (int)S.relpos.size();
        // positions of rows k0.. of d inside t: walk the remaining blocks
        {
          const std::vector<int>& tl = st[sn_last(t)];
          size_t tp = 0;
          for(size_t ii = i; ii < bl.size(); ii++)
          {
            const int q = bl[ii];
            int off;
            if(q >= sn_b0[t] && q < sn_b0[t+1]) off = colstart[q] - S.sn_c0[t];
            else
            {
              while(tp < tl.size() && tl[tp] < q) tp++;
              if(tp >= tl.size() || tl[tp] != q)
                SYM_FAIL("internal error: structure of supernode %d not nested in ancestor %d", d, t);
              off = belowoff[belowoff_ptr[t] + (int)tp];
            }
            for(int a = 0; a < G.w[border[q]]; a++) S.relpos.push_back(off + a);
          }
        }
        S.relpos.push_back(S.sn_rowptr[t+1] - S.sn_rowptr[t] - 1);      // augmented row -> augmented row
        if((int)S.relpos.size() - relbase != nrows_d - k0) SYM_FAIL("internal error: relpos size mismatch");
        if(k0 == wd) { sn_parent[d] = t; S.sn_prel[d] = relbase; }
        if(mf_src)
        {
          // only the map into the parent is needed: the update travels up the tree from there
          while(i < bl.size() && sn_of_b[bl[i]] == t) { krow += G.w[border[bl[i]]]; i++; }
          break;
        }
        // one sub-task per target var-block (another rank's supernode is not a source here)
        while(i < bl.size() && sn_of_b[bl[i]] == t)
        {
          if(mine(d)) subs.push_back({S.sn_level[d], t, bl[i], d, krow, relbase + (krow - k0)});
          krow += G.w[border[bl[i]]];
          i++;
        }
      }
    }
    // levels with many small sources use the two-phase update
    {
      S.upd_syrk.assign(S.nlevels, 0); S.u_off.assign(nsn, -1); S.uscr_size = 0;
      const int syrk_min = env_int("DOGLEG_AMD_SYRK_MIN", 100);
      for(int l = 0; l < S.mf_level0; l++)
      {
        const int n = S.lvl_ptr[l+1] - S.lvl_ptr[l];
        if(syrk_min <= 0 || n < syrk_min) continue;
        bool ok = true;
        for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1] && ok; i++)
        {
          const int d = S.lvl_sn[i];
          const int64_t wd = S.sn_c0[d+1] - S.sn_c0[d], mb = S.sn_rowptr[d+1] - S.sn_rowptr[d] - wd;
          if(mb > 240 || wd <= 8) ok = false;        // <= 15 x 15 tiles of 16 rows: 8 tiles per wave, 16 waves
        }
        if(!ok) continue;
        S.upd_syrk[l] = 1;
        int64_t off = 0;
        for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1]; i++)
        {
          const int d = S.lvl_sn[i];
          const int64_t mb = S.sn_rowptr[d+1] - S.sn_rowptr[d] - (S.sn_c0[d+1] - S.sn_c0[d]);
          S.u_off[d] = off; off += mb*mb;
        }
        S.uscr_size = std::max(S.uscr_size, off);
      }
      // the update matrices of the multifrontal region live until the parent has read them
      S.mf_cptr.assign(nsn + 1, 0);
      for(int d = 0; d < nsn; d++)
        if(S.sn_level[d] >= S.mf_level0)
        {
          const int64_t mb = S.sn_rowptr[d+1] - S.sn_rowptr[d] - (S.sn_c0[d+1] - S.sn_c0[d]);
          S.u_off[d] = S.uscr_size; S.uscr_size += (mb*(mb + 1)/2 + 1) & ~(int64_t)1;     // 16-byte aligned slots
          if(sn_parent[d] >= 0) S.mf_cptr[sn_parent[d] + 1]++;
        }
      for(int t = 0; t < nsn; t++) S.mf_cptr[t+1] += S.mf_cptr[t];
      S.mf_child.resize(S.mf_cptr[nsn]);
      {
        std::vector<int> nx(S.mf_cptr.begin(), S.mf_cptr.end() - 1);
        for(int d = 0; d < nsn; d++)
          if(S.sn_level[d] >= S.mf_level0 && sn_parent[d] >= 0) S.mf_child[nx[sn_parent[d]]++] = d;
      }
      // children in the order of their expected time of arrival, the latest last: a parent adds the early
      // ones while the late one is still at work (k_factor_level, one-launch region).  The estimate: the
      // longest chain below (and including) a supernode, a supernode counting panel rows x columns.
      {
        std::vector<double> chain(nsn, 0.0);
        for(int l = 0; l < S.nlevels; l++)
          for(int i = S.lvl_ptr[l]; i < S.lvl_ptr[l+1]; i++)
          {
            const int t = S.lvl_sn[i];
            double below = 0.0;
            for(int k = S.mf_cptr[t]; k < S.mf_cptr[t+1]; k++) below = std::max(below, chain[S.mf_child[k]]);
            chain[t] = below + (double)(S.sn_rowptr[t+1] - S.sn_rowptr[t])*(S.sn_c0[t+1] - S.sn_c0[t]);
          }
        for(int t = 0; t < nsn; t++)
          std::stable_sort(S.mf_child.begin() + S.mf_cptr[t], S.mf_child.begin() + S.mf_cptr[t+1],
                           [&](int a, int b) { return chain[a] < chain[b]; });
      }
      S.mf_rec.resize(S.mf_child.size());
      for(int t = 0; t < nsn; t++)
      {
        if(S.mf_cptr[t+1] == S.mf_cptr[t]) continue;
        // layout of t's factor workgroup (k_factor_level): panel with an even leading dimension,
        // the update matrix behind it when both (and one scratch double) fit in LDS
        const int w = S.sn_c0[t+1] - S.sn_c0[t], nrows = S.sn_rowptr[t+1] - S.sn_rowptr[t], mb = nrows - w;
        const int ldp = (nrows + 1) & ~1;
        const int jsp = sym_w_split(w, nrows);            // -1: the update matrix stays in HBM
        const bool u_lds = jsp >= 0;
        const int nlin = u_lds ? (int)sym_w_linear(mb, jsp) : 0;
        const int wt_off = u_lds ? ldp*w : 0x8000, trash = ldp*w + nlin;
        for(int k = S.mf_cptr[t]; k < S.mf_cptr[t+1]; k++)
        {
          const int c = S.mf_child[k];
          const int mc = S.sn_rowptr[c+1] - S.sn_rowptr[c] - (S.sn_c0[c+1] - S.sn_c0[c]);
          const int* map = &S.relpos[S.sn_prel[c]];
          const int nc = mc*(mc + 1)/2, npad = (nc + 1023) & ~1023;
          S.mf_rec[k].u_off = S.u_off[c]; S.mf_rec[k].dst_off = (int64_t)S.mf_dst.size();
          S.mf_rec[k].npad = npad; S.mf_rec[k].rsv = 0;
          S.mf_dst.reserve(S.mf_dst.size() + npad);
          for(int j = 0; j < mc; j++)
            for(int i = j; i < mc; i++)
            {
              const int fi = map[i], fj = map[j];
              const int jw = fj - w;
              const int d = (fj < w) ? fi + fj*ldp
                          : (u_lds && jw >= jsp) ? (mb - jw)*ldp + (fi - fj)          // in the top block's upper triangle
                          : wt_off + jw*mb - jw*(jw - 1)/2 + (fi - fj);
              S.mf_dst.push_back((uint16_t)d);
            }
          for(int e = nc; e < npad; e++) S.mf_dst.push_back((uint16_t)trash);
        }
      }
      S.uscr_size += 1024;                 // the padded tail of the last child is read (and dropped)
      S.fw_item.resize(S.fw_sn.size());
      // the work item of every child (persistent top region of the factorisation: its flag); -1: none here
      {
        std::vector<int> sn_item(nsn, -1);
        for(size_t k = 0; k < S.fw_sn.size(); k++) if(sn_item[S.fw_sn[k]] < 0) sn_item[S.fw_sn[k]] = (int)k;
        for(size_t k = 0; k < S.mf_rec.size(); k++) S.mf_rec[k].rsv = sn_item[S.mf_child[k]];
      }
      for(size_t k = 0; k < S.fw_sn.size(); k++)
      {
        const int s2 = S.fw_sn[k];
        FwItem& it = S.fw_item[k];
        it.s = s2; it.r0 = S.fw_r0[k]; it.r1 = S.fw_r1[k]; it.w = S.sn_c0[s2+1] - S.sn_c0[s2];
        it.nrows = S.sn_rowptr[s2+1] - S.sn_rowptr[s2]; it.col0 = S.sn_c0[s2];
        it.bd0 = S.sn_bd_ptr[s2]; it.nbd = S.sn_bd_ptr[s2+1] - S.sn_bd_ptr[s2];
        it.lx = S.sn_lx[s2]; it.top = S.sn_top[s2]; it.u_off = S.u_off[s2];
        it.ch0 = S.mf_cptr[s2]; it.nch = S.mf_cptr[s2+1] - S.mf_cptr[s2];
        it.bdw = 0; it.jsp = sym_w_split(it.w, it.nrows);
        it.rep = 0; it.tj0 = 0; it.tj1 = 1 << 20; it.pad = 0; it.sliced = 0; it.eA = 0; it.eB = 0; it.rsv2 = 0;
        if(it.nbd > 0)
        {
          // members of equal width (the usual case: points): no list lookup in the kernel
          const int* mc = &S.sn_bd_col[it.bd0];
          const int wd0 = (it.nbd > 1 ? mc[1] : it.w) - mc[0];
          bool same = mc[0] == 0;
          for(int m = 0; m < it.nbd && same; m++) if(((m + 1 < it.nbd) ? mc[m+1] : it.w) - mc[m] != wd0) same = false;
          if(same) it.bdw = wd0;
        }
      }
    }
    std::sort(subs.begin(), subs.end(), [](const Sub& a, const Sub& b) {
      if(a.lvl != b.lvl) return a.lvl < b.lvl;
      if(a.t != b.t) return a.t < b.t;
      if(a.q != b.q) return a.q < b.q;
      return a.d < b.d; });
    S.ui_lvl_ptr.assign(S.nlevels + 1, 0);
    S.ui_ptr.push_back(0);
    std::vector<int> item_lvl;
    for(size_t i = 0; i < subs.size(); i++)
    {
      const Sub& u = subs[i];
      const bool newitem = (i == 0) || subs[i-1].lvl != u.lvl || subs[i-1].t != u.t || subs[i-1].q != u.q;
      if(newitem)
      {
        if(i != 0) S.ui_ptr.push_back((int)i);
        S.ui_t.push_back(u.t);
        S.ui_col.push_back(colstart[u.q] - S.sn_c0[u.t]);
        S.ui_nc.push_back(G.w[border[u.q]]);
        S.ui_lvl_ptr[u.lvl + 1]++;
        item_lvl.push_back(u.lvl);
      }
      const int nrows_d = S.sn_rowptr[u.d+1] - S.sn_rowptr[u.d];
      SymSub ss;
      ss.src = S.sn_lx[u.d] + u.ka; ss.rel = u.rel; ss.nrows_d = nrows_d;
      ss.wd = S.sn_c0[u.d+1] - S.sn_c0[u.d]; ss.m = nrows_d - u.ka;
      S.usub.push_back(ss);
      if(S.upd_syrk[u.lvl])
      {
        const int64_t mb = nrows_d - ss.wd, k0 = u.ka - ss.wd;
        S.usub_u.push_back(S.u_off[u.d] + k0*mb - k0*(k0 - 1)/2);      // packed lower triangle: diagonal of column k0
      }
      else S.usub_u.push_back(-1);
    }
    if(!subs.empty()) S.ui_ptr.push_back((int)subs.size());
    for(int l = 0; l < S.nlevels; l++) S.ui_lvl_ptr[l+1] += S.ui_lvl_ptr[l];
    // work units
    S.uw_lvl_ptr.assign(S.nlevels + 1, 0); S.uf_lvl_ptr.assign(S.nlevels + 1, 0);
    S.upart_size = 0;
    const int nitems = (int)S.ui_t.size();
    // chunking by estimated cost: a light sub-task (narrow source) counts 1, a
    // heavy one counts by its thread-iterations; a unit is closed at UNIT_COST
    const int unit_cost = 1024;
    const int unit_cost_gather = std::max(1, unit_cost/64);
    std::vector<int> cuts;
    for(int it = 0; it < nitems; it++)
    {
      const int s0 = S.ui_ptr[it], s1 = S.ui_ptr[it+1];
      const int lvl = item_lvl[it];
      const int t = S.ui_t[it];
      const int64_t slab = (int64_t)(S.sn_rowptr[t+1] - S.sn_rowptr[t])*S.ui_nc[it];
      cuts.clear(); cuts.push_back(s0);
      long acc = 0;
      for(int st = s0; st < s1; st++)
      {
        const SymSub& u = S.usub[st];
        const long c = S.upd_syrk[lvl] ? unit_cost_gather : ((u.wd <= 8) ? 1 : 64 + (long)u.wd*((u.m + 255)/256));
        if(acc > 0 && acc + c > unit_cost) { cuts.push_back(st); acc = 0; }
        acc += c;
      }
      cuts.push_back(s1);
      const int nch = (int)cuts.size() - 1;
      if(nch <= 1)
      {
        S.uw_item.push_back(it); S.uw_s0.push_back(s0); S.uw_s1.push_back(s1); S.uw_part.push_back(-1);
        S.uw_lvl_ptr[lvl + 1]++;
      }
      else
      {
        S.uf_item.push_back(it); S.uf_n.push_back(nch); S.uf_off.push_back(S.upart_size);
        S.uf_lvl_ptr[lvl + 1]++;
        for(int k = 0; k < nch; k++)
        {
          S.uw_item.push_back(it); S.uw_s0.push_back(cuts[k]); S.uw_s1.push_back(cuts[k+1]);
          S.uw_part.push_back(S.upart_size + (int64_t)k*slab);
          S.uw_lvl_ptr[lvl + 1]++;
        }
        S.upart_size += (int64_t)nch*slab;
      }
    }
    for(int l = 0; l < S.nlevels; l++) { S.uw_lvl_ptr[l+1] += S.uw_lvl_ptr[l]; S.uf_lvl_ptr[l+1] += S.uf_lvl_ptr[l]; }
  }

  return 0; };
  auto step9a = [&](char* err, int errlen) -> int {
  // ------------------------------------------- 9a. Jt*x lists (per var-block)
  {
    // inverted index: var-block -> local row-blocks containing it (row order)
    std::vector<int> rptr(nvb + 1, 0);
    for(const RowBlock& b : rbs) if(b.local) for(int x = 0; x < b.nvb; x++) rptr[rb_vb[b.vptr + x] + 1]++;
    for(int v = 0; v < nvb; v++) rptr[v+1] += rptr[v];
    S.oblk.resize(nvb);
    S.contrib.resize(rptr[nvb]);
    std::vector<int> nx(rptr.begin(), rptr.end() - 1);
    for(int bi = 0; bi < nrb; bi++)
    {
      const RowBlock& b = rbs[bi];
      if(!b.local) continue;
      for(int x = 0; x < b.nvb; x++)
      {
        SymContrib c; c.base = b.lbase; c.r0 = b.lr0; c.len = (uint16_t)b.len;
        c.nrows = (uint8_t)b.nrows; c.pad = 0; c.offI = c.offJ = (uint16_t)rb_off[b.vptr + x];
        S.contrib[nx[rb_vb[b.vptr + x]]++] = c;
      }
    }
    for(int v = 0; v < nvb; v++)
    {
      const int q = bpos[v], t = sn_of_b[q];
      const int ld = S.sn_rowptr[t+1] - S.sn_rowptr[t];
      const int lc = colstart[q] - S.sn_c0[t];
      SymOutBlock o; memset(&o, 0, sizeof(o));
      o.dest = S.sn_lx[t] + lc + (int64_t)lc*ld; o.ld = ld; o.var0 = S.vb_start[v];
      o.nI = o.nJ = (uint8_t)G.w[v]; o.diag = 1;
      S.oblk[v] = o;
    }
    S.jtx_nparts = 0; S.jtx_fin_ptr.assign(1, 0); S.jtx_fin_blk.clear();
    S.jtx_covers_all = true;
    for(int v = 0; v < nvb; v++)
    {
      const int c0 = rptr[v], c1 = rptr[v+1];
      if(c1 == c0) { S.jtx_covers_all = false; continue; }
      int chunk = CH_JTX;
      if((c1 - c0 + chunk - 1)/chunk > MAXCH_JTX) chunk = (c1 - c0 + MAXCH_JTX - 1)/MAXCH_JTX;
      const int nch = (c1 - c0 + chunk - 1)/chunk;
      if(nch == 1) S.jtx_task.push_back({v, c0, c1, -1, S.vb_start[v], G.w[v]});
      else
      {
        for(int k = 0; k < nch; k++)
          S.jtx_task.push_back({v, c0 + k*chunk, std::min(c1, c0 + (k+1)*chunk), S.jtx_nparts + k, S.vb_start[v], G.w[v]});
        S.jtx_nparts += nch;
        S.jtx_fin_blk.push_back(v); S.jtx_fin_ptr.push_back(S.jtx_nparts);
      }
    }
  }

  return 0; };
  auto step9b = [&](char* err, int errlen) -> int {
  // ------------------------------------------- 9b. JtJ assembly schedule
  // Column-block centric: a task owns (a group of) the output blocks (I,J) of ONE
  // column block J and walks the row-blocks that contain J in batches: a batch's
  // Jacobian rows are staged in LDS once and feed every block of that column.
  {
    constexpr int ACC_CAP = 256, SLOT_CAP = 32, STAGE_CAP = 512, TASK_BATCHES = 16;
    constexpr int RHO_PER_BATCH = 32, PAIRS_PER_BATCH = 256;
    std::vector<int> rptr(nvb + 1, 0);
    for(const RowBlock& b : rbs) if(b.local) for(int x = 0; x < b.nvb; x++) rptr[rb_vb[b.vptr + x] + 1]++;
    for(int v = 0; v < nvb; v++) rptr[v+1] += rptr[v];
    std::vector<int> rrb(rptr[nvb]), rx(rptr[nvb]);
    {
      std::vector<int> nx(rptr.begin(), rptr.end() - 1);
      for(int bi = 0; bi < nrb; bi++)
      {
        const RowBlock& b = rbs[bi];
        if(!b.local) continue;
        for(int x = 0; x < b.nvb; x++) { const int v = rb_vb[b.vptr + x]; rrb[nx[v]] = bi; rx[nx[v]] = x; nx[v]++; }
      }
    }
    std::vector<int> slot_of(nvb, -1), slot_stamp(nvb, -1);
    struct SlotTmp { int I; int64_t dest; int nI; int diag; };
    std::vector<SlotTmp> slots;
    std::vector<int> grp_of, acc_of;            // per slot: group id, accumulator offset inside the group
    S.asm_part_size = 0;
    // ---- MFMA path.  The row-blocks of a column block J are partitioned into classes of
    // identical row layout (AsmShape); inside a class every block (I,J) is either persistent
    // (same I in every row-block) or transient (all I different).  J qualifies when every
    // transient block receives exactly one contribution overall.
    const bool use_mfma = env_int("DOGLEG_AMD_ASM_MFMA", 1) != 0;
    S.asm_ts_off = env_int("DOGLEG_AMD_ASM_MFMA", 1) == 2;
    const int KG_PER_TASK_T = 64, KG_PER_TASK_P = 256;
    std::map<std::vector<int>, int> shape_ids;
    std::vector<int> tseen(nvb, -1), pseen(nvb, -1), fin_of(nvb, -1);
    struct Ord { int I, nI, offI; bool P; };
    struct Cls { std::vector<int> es; std::vector<Ord> ords; int MP, MT, nP, nT, shape, slot0, acc_size, rider, pq;
                 int lay, xJ; uint64_t mask; };
    // layout id of every row-block: number of blocks, their widths and offsets inside a row
    std::vector<int> rb_layout(nrb, -1);
    {
      std::map<std::vector<int>, int> ids;
      std::vector<int> k, kprev;
      int prev = -1;
      for(int bi = 0; bi < nrb; bi++)
      {
        const RowBlock& b = rbs[bi];
        k.clear();
        for(int x = 0; x < b.nvb; x++) { k.push_back(G.w[rb_vb[b.vptr + x]]); k.push_back(rb_off[b.vptr + x]); }
        if(prev >= 0 && k == kprev) { rb_layout[bi] = prev; continue; }
        auto it = ids.find(k);
        if(it == ids.end()) it = ids.emplace(k, (int)ids.size()).first;
        rb_layout[bi] = prev = it->second; kprev = k;
      }
    }
    std::vector<int> mtask_acc;          // accumulator size of every MFMA task (parallel to asm_mtask)
    std::vector<int> mtask_blk, mtask_rider;   // ... its column block, the rider it carries (-1: none)
    std::vector<Cls> classes;
    std::vector<int> cnt_same, key;
    std::vector<std::vector<int64_t>> fin_lists;
    SYM_TICK("9b.1 riders");
    // riders: dense column blocks whose only output block is their own diagonal
    std::vector<int> rb_host(nrb, -1), rb_rider(nrb, -1);
    std::vector<char> is_rider_blk(nvb, 0);
    std::vector<std::vector<int64_t>> rider_parts(nvb);
    const int RIDER_MIN = env_int("DOGLEG_AMD_RIDER_MIN", 4096);
    if(use_mfma && RIDER_MIN > 0)
      for(int Jr = 0; Jr < nvb; Jr++)
      {
        if(rptr[Jr+1] - rptr[Jr] < RIDER_MIN) continue;
        bool last_everywhere = true;
        for(int e = rptr[Jr]; e < rptr[Jr+1] && last_everywhere; e++)
        {
          const RowBlock& b = rbs[rrb[e]];
          for(int x = 0; x < b.nvb; x++) if(bpos[rb_vb[b.vptr + x]] > bpos[Jr]) { last_everywhere = false; break; }
        }
        if(!last_everywhere) continue;
        is_rider_blk[Jr] = 1;
        for(int e = rptr[Jr]; e < rptr[Jr+1]; e++)
        {
          const int bi = rrb[e];
          const RowBlock& b = rbs[bi];
          int host = -1, best = 0;           // the other block of the row with the longest row list
          for(int x = 0; x < b.nvb; x++)
          {
            const int I = rb_vb[b.vptr + x];
            if(I == Jr) continue;
            const int cnt = rptr[I+1] - rptr[I];
            if(cnt > best || (cnt == best && host >= 0 && bpos[I] < bpos[host])) { best = cnt; host = I; }
          }
          rb_host[bi] = host; rb_rider[bi] = host >= 0 ? Jr : -1;
        }
      }
    auto build_mfma = [&](int J, int r0, int r1, int qJ, int t, int ld, int lc, int nJ, bool is_rider) -> bool
    {
      // 1. classes of identical layout: same row layout id, same position of J in the row, same set
      //    of blocks eliminated after J, same rider
      classes.clear();
      int last_cls = -1;
      for(int e = r0; e < r1; e++)
      {
        const int bi = rrb[e];
        const RowBlock& b = rbs[bi];
        if(is_rider && rb_host[bi] >= 0) continue;             // carried by another column block's tasks
        uint64_t mask = 0;
        if(is_rider) mask = 1;                                  // a rider is the last block of its rows
        else
        {
          if(b.nvb > 64) return false;
          for(int x = 0; x < b.nvb; x++) if(bpos[rb_vb[b.vptr + x]] >= qJ) mask |= 1ull << x;
        }
        const int rider = rb_host[bi] == J ? rb_rider[bi] : -1;
        // a rider only reads its own block: rows with the same offset of that block are one class,
        // whatever else they contain
        const int lay = is_rider ? rb_off[b.vptr + rx[e]] : rb_layout[bi], xJ = is_rider ? 0 : rx[e];
        int c = -1;
        if(last_cls >= 0 && classes[last_cls].lay == lay && classes[last_cls].xJ == xJ &&
           classes[last_cls].mask == mask && classes[last_cls].rider == rider) c = last_cls;
        else
          for(size_t q = 0; q < classes.size(); q++)
            if(classes[q].lay == lay && classes[q].xJ == xJ && classes[q].mask == mask && classes[q].rider == rider) { c = (int)q; break; }
        if(c < 0)
        {
          if(classes.size() >= (is_rider ? 4096u : 64u)) return false;
          c = (int)classes.size(); classes.emplace_back();
          Cls& C = classes.back();
          C.lay = lay; C.xJ = xJ; C.mask = mask; C.rider = rider;
          for(int x = 0; x < b.nvb; x++)
          { const int I = rb_vb[b.vptr + x]; if(bpos[I] >= qJ) C.ords.push_back({I, G.w[I], rb_off[b.vptr + x], true}); }
        }
        classes[c].es.push_back(e); last_cls = c;
      }
      // 2. persistent / transient blocks of every class
      for(Cls& C : classes)
      {
        const int nord = (int)C.ords.size();
        cnt_same.assign(nord, 0);
        for(int e : C.es)
        {
          const RowBlock& b = rbs[rrb[e]];
          int k = 0;
          for(int x = 0; x < b.nvb; x++)
          { const int I = rb_vb[b.vptr + x]; if(bpos[I] < qJ) continue; if(I == C.ords[k].I) cnt_same[k]++; k++; }
        }
        C.MP = C.MT = C.nP = C.nT = 0;
        for(int k = 0; k < nord; k++)
        {
          C.ords[k].P = cnt_same[k] == (int)C.es.size();
          if(C.ords[k].P) { C.MP += C.ords[k].nI; C.nP++; pseen[C.ords[k].I] = J; } else { C.MT += C.ords[k].nI; C.nT++; }
        }
        if(C.MP > 16 || C.MT > 16) return false;
        // window of row columns the kernel stages in LDS
        int lo = rb_off[rbs[rrb[C.es[0]]].vptr + rx[C.es[0]]], hi = lo + nJ;
        for(const Ord& o : C.ords) { lo = std::min(lo, o.offI); hi = std::max(hi, o.offI + o.nI); }
        if(hi - lo > 64 && !is_rider) return false;
        if(hi - lo > 255) return false;
      }
      for(const Cls& C : classes)
      {
        if(C.nT == 0) continue;
        for(int e : C.es)
        {
          const RowBlock& b = rbs[rrb[e]];
          int k = 0;
          for(int x = 0; x < b.nvb; x++)
          {
            const int I = rb_vb[b.vptr + x];
            if(bpos[I] < qJ) continue;
            if(!C.ords[k].P) { if(tseen[I] == J || pseen[I] == J) return false; tseen[I] = J; }
            k++;
          }
        }
      }
      // 3. emit
      const int64_t panel = S.sn_lx[t] + (int64_t)lc*ld;
      const int first_task = (int)S.asm_mtask.size();
      for(Cls& C : classes)
      {
        const int offJ = rb_off[rbs[rrb[C.es[0]]].vptr + rx[C.es[0]]];
        key.assign(1, nJ); key.push_back(offJ);
        for(const Ord& o : C.ords) { key.push_back(o.nI); key.push_back(o.offI); key.push_back(o.P); key.push_back(o.I == C.rider); }
        auto it = shape_ids.find(key);
        if(it != shape_ids.end()) C.shape = it->second;
        else
        {
          AsmShape sh; memset(&sh, 0, sizeof(sh));
          for(int m = 0; m < 16; m++) { sh.pcol[m] = sh.tcol[m] = -1; sh.tj[m] = 0xFF; }
          int mp = 0, mt = 0, ip = 0, jt = 0;
          for(const Ord& o : C.ords)
          {
            for(int a = 0; a < o.nI; a++)
              if(o.P) { sh.pcol[mp] = (int16_t)(o.offI + a); sh.pslot[mp] = (uint8_t)ip; sh.pa[mp] = (uint8_t)a; mp++; }
              else    { sh.tcol[mt] = (int16_t)(o.offI + a); sh.tj[mt] = (uint8_t)jt; sh.ta[mt] = (uint8_t)a; mt++; }
            if(o.P && o.I == C.rider) { sh.offR = (uint16_t)o.offI; sh.nJr = (uint8_t)o.nI; sh.rslot = (uint8_t)ip; }
            if(o.P && o.I == J) sh.dslot = (uint8_t)ip;
            if(o.P) ip++; else jt++;
          }
          { int mp2 = 0, acc = 0;
            for(const Ord& o : C.ords) if(o.P)
            { for(int a = 0; a < o.nI; a++) { sh.paccoff[mp2] = (uint8_t)(acc + a); sh.pnI[mp2] = (uint8_t)o.nI; mp2++; } acc += o.nI*nJ; } }
          int lo = offJ, hi = offJ + nJ;
          for(const Ord& o : C.ords) { lo = std::min(lo, o.offI); hi = std::max(hi, o.offI + o.nI); }
          sh.col0 = (uint16_t)lo; sh.ncopy = (uint8_t)(hi - lo);
          for(int m = 0; m < 16; m++) { if(sh.pcol[m] >= 0) sh.pcol[m] -= lo; if(sh.tcol[m] >= 0) sh.tcol[m] -= lo; }
          if(sh.nJr > 0) sh.offR -= lo;
          S.asm_lds_len = std::max(S.asm_lds_len, ((hi - lo + 15)/16)*16 + 2);
          sh.offJ = (uint16_t)(offJ - lo); sh.nJ = (uint8_t)nJ; sh.MP = (uint8_t)C.MP; sh.MT = (uint8_t)C.MT;
          sh.nT = (uint8_t)C.nT; sh.smax = (uint8_t)std::min(4, 16/nJ);
          C.shape = (int)S.asm_shape.size(); S.asm_shape.push_back(sh); shape_ids[key] = C.shape;
        }
        const int smax = C.nT > 0 ? std::min(4, 16/nJ) : 4;
        const int kg_cap = C.nT > 0 ? KG_PER_TASK_T : KG_PER_TASK_P;
        C.slot0 = (int)S.asm_slot.size(); C.acc_size = 0;
        for(const Ord& o : C.ords) if(o.P)
        {
          AsmSlot sl; sl.dest = slots[slot_of[o.I]].dest; sl.ld = ld; sl.accoff = (uint16_t)C.acc_size;
          sl.nI = (uint8_t)o.nI; sl.diag = (uint8_t)(o.I == J);
          S.asm_slot.push_back(sl); C.acc_size += o.nI*nJ;
        }
        C.pq = (int)S.asm_pdest.size();
        for(const Ord& o : C.ords) if(o.P) S.asm_pdest.push_back((int)(slots[slot_of[o.I]].dest - panel));
        bool task_open = false, kg_open = false;
        int kg_rows = 0, kg_slots = 0, kg_in_task = 0;
        auto close_kg = [&]() {
          if(!kg_open) return;
          AsmKG& g = S.asm_kg.back();
          g.meta |= (uint32_t)kg_slots << 8 | 1u << 11; kg_open = false;
          // at most two transient destinations: they ride in the record
          const int nent = kg_slots*C.nT;
          if(C.nT > 0 && nent <= 2 && g.tq + nent <= (int)S.asm_tdest.size())
          { for(int q = 0; q < nent; q++) g.td[q] = S.asm_tdest[g.tq + q]; g.meta |= 1u << 13; } };
        auto close_task = [&]() { close_kg(); if(task_open) { S.asm_mtask.back().kg1 = (int)S.asm_kg.size(); task_open = false; } };
        auto open_kg = [&]() {
          AsmKG g; g.base[0] = g.base[1] = g.base[2] = g.base[3] = -1; g.tq = (int)S.asm_tdest.size(); g.meta = 0;
          g.xr[0] = g.xr[1] = g.xr[2] = g.xr[3] = 0;
          g.td[0] = g.td[1] = 0;
          S.asm_kg.push_back(g); kg_open = true; kg_rows = 0; kg_slots = 0; kg_in_task++; };
        for(int e : C.es)
        {
          const RowBlock& b = rbs[rrb[e]];
          const bool fits = kg_open && kg_rows + b.nrows <= 4 && kg_slots < smax;
          const int need_kg = b.nrows > 4 ? 2 : (fits ? 0 : 1);
          if(!task_open || kg_in_task + need_kg > kg_cap)
          {
            close_task();
            AsmMTask T; T.kg0 = (int)S.asm_kg.size(); T.kg1 = -1; T.slot0 = C.slot0; T.shape = C.shape; T.ld = ld;
            T.pq = C.pq; T.panel = panel; T.part = -1; T.rpart = -1; T.jvar = -1; T.pad = 0;
            mtask_acc.push_back(C.acc_size); mtask_blk.push_back(J); mtask_rider.push_back(C.rider);
            if(C.rider >= 0)
            {
              const int nr = G.w[C.rider];
              T.rpart = S.asm_part_size; S.asm_part_size += nr*nr;
              rider_parts[C.rider].push_back(T.rpart);
            }
            S.asm_mtask.push_back(T); task_open = true; kg_in_task = 0;
          }
          const int base = b.lbase;
          if(b.nrows > 4)
          {
            close_kg();
            const int tq = (int)S.asm_tdest.size();
            for(int h = 0; h < 2; h++)
            {
              open_kg(); S.asm_kg.back().tq = tq;
              for(int r = 4*h; r < std::min(b.nrows, 4*h + 4); r++)
              { S.asm_kg.back().base[r - 4*h] = base + r*b.len; S.asm_kg.back().xr[r - 4*h] = b.lr0 + r; }
              kg_slots = 1;
              if(h == 0) { S.asm_kg.back().meta |= 1u << 8; kg_open = false; }    // no store yet: the block continues
              else close_kg();
            }
          }
          else
          {
            if(!(kg_open && kg_rows + b.nrows <= 4 && kg_slots < smax)) { close_kg(); open_kg(); }
            for(int r = 0; r < b.nrows; r++)
            {
              S.asm_kg.back().base[kg_rows] = base + r*b.len; S.asm_kg.back().xr[kg_rows] = b.lr0 + r;
              S.asm_kg.back().meta |= (uint32_t)kg_slots << (2*kg_rows);
              kg_rows++;
            }
            kg_slots++;
          }
          if(C.nT > 0)
          {
            int k = 0;
            for(int x = 0; x < b.nvb; x++)
            {
              const int I = rb_vb[b.vptr + x];
              if(bpos[I] < qJ) continue;
              if(!C.ords[k].P) S.asm_tdest.push_back((int)(slots[slot_of[I]].dest - panel));
              k++;
            }
          }
        }
        close_task();
      }
      // 4. several tasks: persistent blocks go through partials, summed per destination block
      const int ntask = (int)S.asm_mtask.size() - first_task;
      const bool carried = is_rider && !rider_parts[J].empty();
      if(ntask > 1 || carried)
      {
        const int fin0 = (int)S.asm_fin2.size();
        fin_lists.clear();
        for(int k = first_task; k < first_task + ntask; k++)
        {
          AsmMTask& T = S.asm_mtask[k];
          T.part = S.asm_part_size; S.asm_part_size += mtask_acc[k];
        }
        for(const Cls& C : classes)
        {
          int sidx = 0;
          for(const Ord& o : C.ords) if(o.P)
          {
            const AsmSlot& sl = S.asm_slot[C.slot0 + sidx]; sidx++;
            int f;
            if(fin_of[o.I] >= fin0 && fin_of[o.I] < (int)S.asm_fin2.size() && S.asm_fin2[fin_of[o.I]].dest == sl.dest) f = fin_of[o.I];
            else
            {
              f = (int)S.asm_fin2.size(); fin_of[o.I] = f;
              AsmFin2 F; F.dest = sl.dest; F.ld = ld; F.list0 = 0; F.nlist = 0; F.nI = sl.nI; F.nJ = (uint8_t)nJ;
              F.diag = sl.diag; F.to_part = 0;
              S.asm_fin2.push_back(F); fin_lists.emplace_back();
            }
            for(int k = first_task; k < first_task + ntask; k++)
            {
              const AsmMTask& T = S.asm_mtask[k];
              if(T.slot0 == C.slot0) fin_lists[f - fin0].push_back(T.part + sl.accoff);
            }
          }
        }
        if(carried)
        {
          if(fin_lists.empty())          // every row of the rider is carried by other tasks
          {
            AsmFin2 F; F.dest = S.sn_lx[t] + lc + (int64_t)lc*ld; F.ld = ld; F.list0 = 0; F.nlist = 0;
            F.nI = F.nJ = (uint8_t)nJ; F.diag = 1; F.to_part = 0;
            S.asm_fin2.push_back(F); fin_lists.emplace_back();
          }
          fin_lists[0].insert(fin_lists[0].begin(), rider_parts[J].begin(), rider_parts[J].end());
        }
        for(size_t f = 0; f < fin_lists.size(); f++)
        {
          S.asm_fin2[fin0 + f].list0 = (int)S.asm_fin2_list.size();
          S.asm_fin2[fin0 + f].nlist = (int)fin_lists[f].size();
          S.asm_fin2_list.insert(S.asm_fin2_list.end(), fin_lists[f].begin(), fin_lists[f].end());
        }
      }
      return true;
    };
    SYM_TICK("9b.2 column blocks");
    for(int pass = 0; pass < 2; pass++)          // pass 1: the riders (rows nobody carried + the carried partials)
    for(int J = 0; J < nvb; J++)
    {
      const int r0 = rptr[J], r1 = rptr[J+1];
      if(r0 == r1 || (int)is_rider_blk[J] != pass) continue;
      const int qJ = bpos[J], t = sn_of_b[qJ];
      const int ld = S.sn_rowptr[t+1] - S.sn_rowptr[t];
      const int lc = colstart[qJ] - S.sn_c0[t];
      const int nJ = G.w[J];
      // distinct I's with pos(I) >= pos(J), in first-seen order
      slots.clear();
      for(int e = r0; e < r1; e++)
      {
        const RowBlock& b = rbs[rrb[e]];
        for(int x = 0; x < b.nvb; x++)
        {
          const int I = rb_vb[b.vptr + x];
          if(bpos[I] < qJ || slot_stamp[I] == J) continue;
          slot_stamp[I] = J; slot_of[I] = (int)slots.size();
          int64_t dest;
          if(I == J) dest = S.sn_lx[t] + lc + (int64_t)lc*ld;
          else
          {
            const int ro = rowoff_in(t, bpos[I]);
            if(ro < 0) SYM_FAIL("internal error: JtJ block (%d,%d) missing from the factor structure", bpos[I], qJ);
            dest = S.sn_lx[t] + ro + (int64_t)lc*ld;
          }
          slots.push_back({I, dest, G.w[I], I == J ? 1 : 0});
        }
      }
      if(use_mfma && build_mfma(J, r0, r1, qJ, t, ld, lc, nJ, pass == 1)) continue;
      if(pass == 1) SYM_FAIL("internal error: no assembly schedule for dense column block %d", J);
      // this block falls back to the LDS kernel: the riders it would have carried keep their own tasks
      for(int e = r0; e < r1; e++) if(rb_host[rrb[e]] == J) { rb_host[rrb[e]] = -2; rb_rider[rrb[e]] = -1; }
      // groups of slots that fit the LDS accumulator
      grp_of.assign(slots.size(), 0); acc_of.assign(slots.size(), 0);
      int ngrp = 0;
      {
        int acc = 0, cnt = 0;
        for(size_t k = 0; k < slots.size(); k++)
        {
          const int sz = slots[k].nI*nJ;
          if(cnt > 0 && (acc + sz > ACC_CAP || cnt >= SLOT_CAP)) { ngrp++; acc = 0; cnt = 0; }
          grp_of[k] = ngrp; acc_of[k] = acc; acc += sz; cnt++;
        }
        ngrp++;
      }
      for(int g = 0; g < ngrp; g++)
      {
        // slot table of the group
        const int slot0 = (int)S.asm_slot.size();
        int acc_size = 0, nslots = 0;
        for(size_t k = 0; k < slots.size(); k++) if(grp_of[k] == g)
        {
          AsmSlot sl; sl.dest = slots[k].dest; sl.ld = ld; sl.accoff = (uint16_t)acc_of[k];
          sl.nI = (uint8_t)slots[k].nI; sl.diag = (uint8_t)slots[k].diag;
          S.asm_slot.push_back(sl); nslots++; acc_size = acc_of[k] + slots[k].nI*nJ;
        }
        // rho records with at least one pair in this group, batched by staged size
        const int first_task = (int)S.asm_ctask.size();
        bool task_open = false, batch_open = false;
        std::vector<int> seq0, seq;
        int nb_in_task = 0, stage_used = 0, rho_in_batch = 0, pairs_in_batch = 0;
        auto close_batch = [&]() { if(batch_open) { S.asm_batch.back().rho1 = (int)S.asm_rho.size(); batch_open = false; } };
        auto close_task = [&]() {
          close_batch();
          if(task_open) { S.asm_ctask.back().batch1 = (int)S.asm_batch.size(); task_open = false; } };
        for(int e = r0; e < r1; e++)
        {
          const RowBlock& b = rbs[rrb[e]];
          int npairs = 0;
          for(int x = 0; x < b.nvb; x++)
          { const int I = rb_vb[b.vptr + x]; if(bpos[I] >= qJ && grp_of[slot_of[I]] == g) npairs++; }
          if(npairs == 0) continue;
          const int cnt = b.nrows*b.len;
          const bool direct = cnt > STAGE_CAP;
          const int need = direct ? 0 : cnt;
          if(batch_open && (stage_used + need > STAGE_CAP || rho_in_batch >= RHO_PER_BATCH ||
                            pairs_in_batch + npairs > PAIRS_PER_BATCH)) close_batch();
          if(!batch_open)
          {
            if(!task_open || nb_in_task >= TASK_BATCHES)
            {
              close_task();
              AsmTask T; T.batch0 = (int)S.asm_batch.size(); T.batch1 = -1; T.slot0 = slot0;
              T.nslots = (uint16_t)nslots; T.nJ = (uint8_t)nJ; T.pad = 0; T.acc_size = acc_size; T.pad2 = 0; T.part = -1;
              T.pad = 1;                       // uniform until a row-block with another block sequence shows up
              S.asm_ctask.push_back(T); task_open = true; nb_in_task = 0; seq0.clear();
            }
            S.asm_batch.push_back({(int)S.asm_rho.size(), -1}); batch_open = true; stage_used = 0; nb_in_task++;
            rho_in_batch = 0; pairs_in_batch = 0;
          }
          AsmRho R; R.base = b.lbase; R.pair0 = (int)S.asm_pair.size(); R.len = (uint16_t)b.len;
          R.offJ = (uint16_t)rb_off[b.vptr + rx[e]]; R.stage_off = direct ? (uint16_t)0xFFFF : (uint16_t)stage_used;
          R.nrows = (uint8_t)b.nrows; R.pad = 0;
          S.asm_rho.push_back(R);
          stage_used += need; rho_in_batch++; pairs_in_batch += npairs;
          seq.clear();
          for(int x = 0; x < b.nvb; x++)
          { const int I = rb_vb[b.vptr + x]; if(bpos[I] >= qJ && grp_of[slot_of[I]] == g) seq.push_back(G.w[I]); }
          if(seq0.empty()) { seq0 = seq; int tot = 0; for(int q : seq) tot += q*nJ; if(tot > 128) S.asm_ctask.back().pad = 0; }
          else if(seq != seq0) S.asm_ctask.back().pad = 0;
          if(npairs > PAIRS_PER_BATCH) SYM_FAIL("a measurement row touches too many variable blocks (%d)", npairs);
          for(int x = 0; x < b.nvb; x++)
          {
            const int I = rb_vb[b.vptr + x];
            if(bpos[I] < qJ || grp_of[slot_of[I]] != g) continue;
            AsmPair P; P.offI = (uint16_t)rb_off[b.vptr + x];
            P.acc_nI = (uint16_t)(acc_of[slot_of[I]] | ((G.w[I] - 1) << 12));
            S.asm_pair.push_back(P);
          }
        }
        close_task();
        // multi-task groups: partial accumulators + finalize entry
        const int ntask = (int)S.asm_ctask.size() - first_task;
        if(ntask > 1)
        {
          AsmFin F; F.slot0 = slot0; F.nslots = nslots; F.acc_size = acc_size; F.nparts = ntask;
          F.part0 = S.asm_part_size; F.nJ = nJ;
          S.asm_cfin.push_back(F);
          for(int k = 0; k < ntask; k++) S.asm_ctask[first_task + k].part = S.asm_part_size + (int64_t)k*acc_size;
          S.asm_part_size += (int64_t)ntask*acc_size;
        }
      }
    }
    SYM_TICK("9b.3 sort tasks, runs");
    // group the MFMA tasks by shape (their k-groups move with them) and cut the sequence into
    // runs: a wave loads the shape's lane constants once and streams the k-groups of several
    // small tasks back to back
    {
      const int nt = (int)S.asm_mtask.size();
      std::vector<int> order(nt);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return S.asm_mtask[a].shape < S.asm_mtask[b].shape; });
      // (the runs are cut first -- consecutive tasks of one shape, at most RUN_KG k-groups --, and every run's k-groups are
      // padded with empty ones to a multiple of the kernel's unroll: the kernel then never looks past the end of its run
      // inside an iteration (four scalar compares and a select a k-group).  An empty k-group behind a task's last one has
      // no rows, no slots and no end: it costs a product of zeros.)
      const int RUN_KG = 32;
      std::vector<AsmMTask> tasks2; tasks2.reserve(nt);
      std::vector<AsmKG> kg2; kg2.reserve(S.asm_kg.size() + S.asm_kg.size()/8);
      S.asm_run.clear();
      for(int q = 0; q < nt; )
      {
        AsmRun R; R.task0 = q; R.kg0 = (int)kg2.size();
        const int shape0 = S.asm_mtask[order[q]].shape;
        int nkg = 0;
        do
        {
          AsmMTask T = S.asm_mtask[order[q]];
          const int k0 = (int)kg2.size();
          kg2.insert(kg2.end(), S.asm_kg.begin() + T.kg0, S.asm_kg.begin() + T.kg1);
          kg2.back().meta |= 1u << 12;             // last k-group of its task
          nkg += T.kg1 - T.kg0;
          T.kg0 = k0; T.kg1 = (int)kg2.size();
          tasks2.push_back(T);
          q++;
        }
        while(q < nt && S.asm_mtask[order[q]].shape == shape0 &&
              nkg + (S.asm_mtask[order[q]].kg1 - S.asm_mtask[order[q]].kg0) <= RUN_KG);
        while(((int)kg2.size() - R.kg0) % ASM_KG_ALIGN != 0)
        {
          AsmKG g; g.base[0] = g.base[1] = g.base[2] = g.base[3] = -1; g.tq = 0; g.meta = 1u << 13;      // (no slots; "destinations in the record": none)
          g.xr[0] = g.xr[1] = g.xr[2] = g.xr[3] = 0; g.td[0] = g.td[1] = 0;
          kg2.push_back(g);
        }
        R.task1 = q; R.kg1 = (int)kg2.size();
        S.asm_run.push_back(R);
      }
      S.asm_mtask.swap(tasks2); S.asm_kg.swap(kg2);
      // Jt*x beside JtJ: which task records every var-block sums (task order)
      {
        std::vector<std::vector<int>> ent(nvb);
        for(int kk = 0; kk < nt; kk++)
        {
          const int k = order[kk], J = mtask_blk[k], R = mtask_rider[k];
          ent[J].push_back(16*kk);
          if(R >= 0) ent[R].push_back(16*kk + G.w[J]);
        }
        S.asm_jtx_ok = use_mfma && S.asm_ctask.empty() && nt > 0;
        S.jf_ptr.assign(1, 0); S.jf_ent.clear(); S.jf_var0.resize(nvb); S.jf_w.resize(nvb);
        S.jf_short.clear(); S.jf_long.clear();
        for(int v = 0; v < nvb; v++)
        {
          S.jf_ent.insert(S.jf_ent.end(), ent[v].begin(), ent[v].end());
          S.jf_ptr.push_back((int)S.jf_ent.size());
          S.jf_var0[v] = S.vb_start[v]; S.jf_w[v] = G.w[v];
          if(G.w[v] > 16 || (ent[v].empty() && rptr[v+1] > rptr[v])) S.asm_jtx_ok = false;
          if(ent[v].empty()) continue;                 // no rows: its entries of Jt*x stay zero
          // a single record that is the block's own task (not a ride on another block's): the kernel
          // writes Jt*x itself
          if(ent[v].size() == 1 && (ent[v][0] & 15) == 0) { S.asm_mtask[ent[v][0] >> 4].jvar = S.vb_start[v]; continue; }
          if(ent[v].size() <= 64) S.jf_short.push_back(v); else S.jf_long.push_back(v);
        }
      }
      // persistent destinations, 16 per task in task order (the kernel prefetches them by task index)
      {
        std::vector<int> pd2((size_t)nt*16, 0);
        for(int k = 0; k < nt; k++)
          for(int q = 0; q < 16 && S.asm_mtask[k].pq + q < (int)S.asm_pdest.size(); q++)
            pd2[(size_t)k*16 + q] = S.asm_pdest[S.asm_mtask[k].pq + q];
        S.asm_pdest.swap(pd2);
      }
      // (do all k-groups that store transient blocks carry the destinations themselves?)
      S.asm_td_inline = true;
      for(const AsmMTask& T : S.asm_mtask)
        if(S.asm_shape[T.shape].MT > 0)
          for(int g = T.kg0; g < T.kg1; g++)
            if(((S.asm_kg[g].meta >> 8) & 7) > 0 && (!(S.asm_kg[g].meta & (1u << 13)) || !(S.asm_kg[g].meta & (1u << 11)))) S.asm_td_inline = false;     // (... and stores them: no row-block of more than four rows)
      // (Tried: an XCD-aware launch order -- runs sorted by the position in J of the first row they
      // read, the sorted list cut into 8 segments laid out on the workgroups s, s + 8, ..., so that the
      // two readers of a stretch of J, its points' tasks and its cameras' tasks, meet in one XCD's L2.
      // Measured slower: 120.7 us against 112.6 on config #4, 1.17 ms against 1.07 on config #5 -- the
      // kernel is bound by instruction issue around the K = 4 MFMAs, not by the second pass over J, and
      // the shape-sorted order balances the long camera runs better.)
    }
    SYM_TICK("9b.4 partial-sum stages");
    // long lists are summed hierarchically: chunks of 64 partials -> intermediate partials.
    // Stages run in order; inside a stage the entries with <= 64 partials come first
    // (one wave each), then the longer ones (one workgroup each).
    {
      std::vector<AsmFin2> cur; cur.swap(S.asm_fin2);
      std::vector<int64_t> lst; lst.swap(S.asm_fin2_list);
      std::vector<std::vector<AsmFin2>> stages;
      std::vector<AsmFin2> inter;
      for(int depth = 0; depth < 8; depth++)
      {
        inter.clear();
        for(AsmFin2& F : cur)
        {
          if(F.nlist <= 256) continue;
          const int e = F.nI*F.nJ, nch = (F.nlist + 63)/64;
          const int new0 = (int)lst.size();
          std::vector<int64_t> news;
          for(int c = 0; c < nch; c++)
          {
            AsmFin2 X = F; X.to_part = 1; X.dest = S.asm_part_size; S.asm_part_size += e;
            X.list0 = F.list0 + 64*c; X.nlist = std::min(64, F.nlist - 64*c);
            inter.push_back(X); news.push_back(X.dest);
          }
          lst.insert(lst.end(), news.begin(), news.end());
          F.list0 = new0; F.nlist = nch;
        }
        stages.push_back(cur);
        if(inter.empty()) break;
        cur = inter;
        // the intermediates just created must run BEFORE the entries that consume them
      }
      // stages were collected consumer-first: emit them in reverse
      for(int sidx = (int)stages.size() - 1; sidx >= 0; sidx--)
      {
        std::vector<AsmFin2>& st = stages[sidx];
        std::stable_partition(st.begin(), st.end(), [](const AsmFin2& f) { return f.nlist <= 64; });
        int ns = 0; for(const AsmFin2& f : st) if(f.nlist <= 64) ns++;
        S.fin2_stage.push_back((int)S.asm_fin2.size()); S.fin2_stage.push_back(ns); S.fin2_stage.push_back((int)st.size() - ns);
        S.asm_fin2.insert(S.asm_fin2.end(), st.begin(), st.end());
      }
      S.asm_fin2_list.swap(lst);
    }
    SYM_TICK("9b.5 sentinel");
    // sentinel so that rho[i+1].pair0 closes the pair list of the last rho
    AsmRho R; memset(&R, 0, sizeof(R)); R.pair0 = (int)S.asm_pair.size(); S.asm_rho.push_back(R);
  }

  return 0; };
  auto step10 = [&](char* err, int errlen) -> int {
  // --------------------------------------- 10. forward-solve gather lists
  {
    S.rl_ptr.assign(N + 1, 0);
    for(int d = 0; d < nsn; d++)
    {
      const int wd = S.sn_c0[d+1] - S.sn_c0[d];
      for(int k = S.sn_rowptr[d] + wd; k < S.sn_rowptr[d+1] - 1; k++) S.rl_ptr[S.sn_rows[k] + 1]++;
    }
    for(int k = 0; k < N; k++) S.rl_ptr[k+1] += S.rl_ptr[k];
    S.rl_pos.resize(S.rl_ptr[N]);
    std::vector<int> nx(S.rl_ptr.begin(), S.rl_ptr.end() - 1);
    for(int d = 0; d < nsn; d++)
    {
      const int wd = S.sn_c0[d+1] - S.sn_c0[d];
      for(int k = S.sn_rowptr[d] + wd, j = 0; k < S.sn_rowptr[d+1] - 1; k++, j++)
        S.rl_pos[nx[S.sn_rows[k]]++] = S.sn_scr[d] + j;
    }
  }
  return 0; };
  SYM_TICK("8-10 schedules (4 threads)");
  {
    char e8[256] = "", e9a[256] = "", e9b[256] = "", e10[256] = "";
    int r8 = 0, r9a = 0, r9b = 0, r10 = 0;
    if(true)
    {
      sym_parallel = true;
      std::thread t8([&] { r8 = step8(e8, sizeof(e8)); });
      std::thread t9a([&] { r9a = step9a(e9a, sizeof(e9a)); });
      std::thread t10([&] { r10 = step10(e10, sizeof(e10)); });
      r9b = step9b(e9b, sizeof(e9b));
      t8.join(); t9a.join(); t10.join();
      sym_parallel = false;
    }
    else { r8 = step8(e8, sizeof(e8)); r9a = step9a(e9a, sizeof(e9a)); r9b = step9b(e9b, sizeof(e9b)); r10 = step10(e10, sizeof(e10)); }
    const char* em = r8 ? e8 : (r9a ? e9a : (r9b ? e9b : (r10 ? e10 : nullptr)));
    if(em) SYM_FAIL("%s", em);
  }
  SYM_TICK("done");
  return 0;
}
