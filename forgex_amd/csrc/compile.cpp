// See compile.hpp / program.h.
#include "compile.hpp"

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>

using namespace fxfe;

namespace fxc {
namespace {

using Bits = std::vector<uint64_t>;
inline bool bt(const Bits& b, int i) { return (b[static_cast<size_t>(i) >> 6] >> (i & 63)) & 1u; }
inline void bs(Bits& b, int i) { b[static_cast<size_t>(i) >> 6] |= (uint64_t(1) << (i & 63)); }
inline void bor(Bits& a, const Bits& b) {
   for (size_t i = 0; i < a.size(); ++i) a[i] |= b[i];
}
inline bool bany(const Bits& b) {
   for (uint64_t w : b)
      if (w) return true;
   return false;
}

struct FlatTra {
   int src, dst;
   std::vector<int> acc;   // interval ids whose code points this transition consumes
};

struct Dfa {
   int n = 0;                 // number of states
   int ncol = 0;              // number of columns (intervals, later classes)
   std::vector<int> T;        // [n][ncol]
   std::vector<uint8_t> out;  // accept (A) / hit (R)
   int init = 0;
   bool ok = true;
};

struct IntVecHash {
   size_t operator()(const std::vector<int>& v) const {
      uint64_t h = 1469598103934665603ull;
      for (int x : v) h = (h ^ static_cast<uint32_t>(x)) * 1099511628211ull;
      return static_cast<size_t>(h ^ (h >> 29));
   }
};

// Moore partition refinement; keeps `keep0` (the dead state of A) as state 0 when >= 0.
// `labels` (optional) refines the initial partition beyond `out`; `old2new` (optional) receives the state renumbering (-1 = dropped).
void minimise(Dfa& d, int keep0, const std::vector<int>* labels = nullptr, std::vector<int>* old2new = nullptr) {
   int n = d.n, nc = d.ncol;
   std::vector<int> block(static_cast<size_t>(n));
   for (int s = 0; s < n; ++s) block[static_cast<size_t>(s)] = d.out[static_cast<size_t>(s)] + (labels ? 2 * (*labels)[static_cast<size_t>(s)] : 0);
   int nblocks = 0;
   {
      std::map<int, int> m;
      for (int s = 0; s < n; ++s) {
         auto it = m.find(block[static_cast<size_t>(s)]);
         if (it == m.end()) it = m.emplace(block[static_cast<size_t>(s)], static_cast<int>(m.size())).first;
         block[static_cast<size_t>(s)] = it->second;
      }
      nblocks = static_cast<int>(m.size());
   }
   // every round: states with the same (block, blocks of the successors) signature share a new block, numbered in order of first
   // appearance.  The signatures live in one flat array; a hash finds the earlier states to compare with (exact comparison decides).
   // columns that are equal in every state say the same thing in every signature: one representative each takes part
   std::vector<int> cols;
   {
      std::unordered_multimap<uint64_t, int> col_by_hash;
      for (int c = 0; c < nc; ++c) {
         uint64_t hsh = 1469598103934665603ull;
         for (int s = 0; s < n; ++s) hsh = (hsh ^ static_cast<uint32_t>(d.T[static_cast<size_t>(s) * nc + c])) * 1099511628211ull;
         bool dup = false;
         auto range = col_by_hash.equal_range(hsh);
         for (auto it = range.first; it != range.second && !dup; ++it) {
            const int c2 = it->second;
            bool eq = true;
            for (int s = 0; s < n && eq; ++s) eq = d.T[static_cast<size_t>(s) * nc + c] == d.T[static_cast<size_t>(s) * nc + c2];
            dup = eq;
         }
         if (!dup) {
            col_by_hash.emplace(hsh, c);
            cols.push_back(c);
         }
      }
   }
   const size_t w = cols.size() + 1;
   std::vector<int> sigs(static_cast<size_t>(n) * w);
   std::vector<int> nb(static_cast<size_t>(n));
   std::vector<int> first_of;   // new block -> the first state that has its signature
   std::unordered_multimap<uint64_t, int> by_hash;   // signature hash -> new block
   while (true) {
      first_of.clear();
      by_hash.clear();
      by_hash.reserve(static_cast<size_t>(n) * 2);
      for (int s = 0; s < n; ++s) {
         int* sig = &sigs[static_cast<size_t>(s) * w];
         sig[0] = block[static_cast<size_t>(s)];
         uint64_t hsh = (1469598103934665603ull ^ static_cast<uint32_t>(sig[0])) * 1099511628211ull;
         const int* row = &d.T[static_cast<size_t>(s) * nc];
         for (size_t ci = 0; ci < cols.size(); ++ci) {
            const int b = block[static_cast<size_t>(row[cols[ci]])];
            sig[ci + 1] = b;
            hsh = (hsh ^ static_cast<uint32_t>(b)) * 1099511628211ull;
         }
         int id = -1;
         auto range = by_hash.equal_range(hsh);
         for (auto it = range.first; it != range.second; ++it)
            if (std::memcmp(&sigs[static_cast<size_t>(first_of[static_cast<size_t>(it->second)]) * w], sig, w * sizeof(int)) == 0) {
               id = it->second;
               break;
            }
         if (id < 0) {
            id = static_cast<int>(first_of.size());
            first_of.push_back(s);
            by_hash.emplace(hsh, id);
         }
         nb[static_cast<size_t>(s)] = id;
      }
      int k = static_cast<int>(first_of.size());
      block.swap(nb);
      if (k == nblocks) break;
      nblocks = k;
   }
   // renumber: dead block first (if any), then non-output blocks, then output blocks -- only REACHABLE blocks kept
   std::vector<int> rep(static_cast<size_t>(nblocks), -1);
   for (int s = 0; s < n; ++s)
      if (rep[static_cast<size_t>(block[static_cast<size_t>(s)])] < 0) rep[static_cast<size_t>(block[static_cast<size_t>(s)])] = s;
   std::vector<uint8_t> reach(static_cast<size_t>(nblocks), 0);
   std::vector<int> stack{block[static_cast<size_t>(d.init)]};
   reach[static_cast<size_t>(stack[0])] = 1;
   if (keep0 >= 0) reach[static_cast<size_t>(block[static_cast<size_t>(keep0)])] = 1;
   while (!stack.empty()) {
      int b = stack.back();
      stack.pop_back();
      int s = rep[static_cast<size_t>(b)];
      for (int c = 0; c < nc; ++c) {
         int nb2 = block[static_cast<size_t>(d.T[static_cast<size_t>(s) * nc + c])];
         if (!reach[static_cast<size_t>(nb2)]) {
            reach[static_cast<size_t>(nb2)] = 1;
            stack.push_back(nb2);
         }
      }
   }
   std::vector<int> newid(static_cast<size_t>(nblocks), -1);
   int next = 0;
   if (keep0 >= 0) newid[static_cast<size_t>(block[static_cast<size_t>(keep0)])] = next++;
   for (int pass = 0; pass < 2; ++pass)
      for (int b = 0; b < nblocks; ++b)
         if (reach[static_cast<size_t>(b)] && newid[static_cast<size_t>(b)] < 0 && d.out[static_cast<size_t>(rep[static_cast<size_t>(b)])] == pass)
            newid[static_cast<size_t>(b)] = next++;
   Dfa r;
   r.n = next;
   r.ncol = nc;
   r.T.assign(static_cast<size_t>(next) * nc, 0);
   r.out.assign(static_cast<size_t>(next), 0);
   for (int b = 0; b < nblocks; ++b) {
      int id = newid[static_cast<size_t>(b)];
      if (id < 0) continue;
      int s = rep[static_cast<size_t>(b)];
      r.out[static_cast<size_t>(id)] = d.out[static_cast<size_t>(s)];
      for (int c = 0; c < nc; ++c) r.T[static_cast<size_t>(id) * nc + c] = newid[static_cast<size_t>(block[static_cast<size_t>(d.T[static_cast<size_t>(s) * nc + c])])];
   }
   r.init = newid[static_cast<size_t>(block[static_cast<size_t>(d.init)])];
   if (old2new) {
      old2new->assign(static_cast<size_t>(n), -1);
      for (int s = 0; s < n; ++s) (*old2new)[static_cast<size_t>(s)] = newid[static_cast<size_t>(block[static_cast<size_t>(s)])];
   }
   d = r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Byte-level automata: a class-level DFA composed with the UTF-8 decoder, so that the device walks raw bytes.
// The decoder is the reference's arithmetic one (utf8_m.f90:338-430 `ichar_utf8`): EVERY structurally valid sequence (lead
// byte C0..F7 followed by its 1..3 continuation bytes) yields the code point its payload bits spell, overlong forms and
// values beyond U+10FFFF included.  Structure errors (utf8_m.f90:168-246: a continuation byte without a lead, a lead without
// its continuation bytes, F8..FF) are NOT modelled here: they lead to one absorbing INVALID state and the row is redone by
// the decode pass.
// ---------------------------------------------------------------------------------------------------------------------------
using Sig = std::vector<std::pair<uint32_t, int>>;   // piecewise-constant class function: ascending (start, class), first start 0

void sig_normalise(Sig& s) {
   Sig o;
   for (const auto& pr : s) {
      if (!o.empty() && o.back().first == pr.first) o.pop_back();   // an empty piece: the later one wins
      if (!o.empty() && o.back().second == pr.second) continue;
      o.push_back(pr);
   }
   s.swap(o);
}
int sig_eval(const Sig& s, uint32_t x) {
   int c = s[0].second;
   for (const auto& pr : s) {
      if (pr.first > x) break;
      c = pr.second;
   }
   return c;
}
// the function on [lo, hi) re-based to 0
Sig sig_slice(const Sig& s, uint32_t lo, uint32_t hi) {
   Sig o;
   o.emplace_back(0u, sig_eval(s, lo));
   for (const auto& pr : s)
      if (pr.first > lo && pr.first < hi) o.emplace_back(pr.first - lo, pr.second);
   sig_normalise(o);
   return o;
}
// g'(h) = g(64 h + v) for h < limit
Sig sig_shrink(const Sig& s, uint32_t v, uint32_t limit) {
   Sig o;
   for (const auto& pr : s) {
      const uint32_t ns = pr.first <= v ? 0u : (pr.first - v + 63u) / 64u;
      if (ns >= limit) break;
      o.emplace_back(ns, pr.second);
   }
   sig_normalise(o);
   return o;
}

// the class function with everything outside [lo, hi) marked -1 ("not encodable at this length")
Sig sig_restrict(const Sig& full, uint32_t lo, uint32_t hi) {
   Sig o;
   if (lo > 0) o.emplace_back(0u, -1);
   o.emplace_back(lo, sig_eval(full, lo));
   for (const auto& pr : full)
      if (pr.first > lo && pr.first < hi) o.push_back(pr);
   o.emplace_back(hi, -1);
   sig_normalise(o);
   return o;
}
bool sig_all_invalid(const Sig& sg) { return sg.size() == 1 && sg[0].second < 0; }

struct ByteDfa {
   Dfa d;                  // ncol = 256 until the byte classes are merged
   std::vector<int> fin;   // verdict at the end of a row: 0 / 1, 2 = redo by the decode pass (inside a character, INVALID)
   int inv = -1;           // the INVALID state (-1: none; structure errors lead to the dead state 0 instead)
   bool ok = false;
};

struct ClassDfaView {
   int n;
   std::function<int(int, int)> T;   // (state, class) -> state
   const std::vector<uint8_t>* out;
   int init;
};

constexpr int kMaxByteStates = 4096;

// forward: (q, node); node 0 = between characters, else (continuation bytes still expected, class function of what they can spell)
ByteDfa build_forward_bytes(const ClassDfaView& A, const Sig& full, const std::vector<int>& fin_class, bool invalid_sink) {
   ByteDfa r;
   std::vector<std::pair<int, Sig>> nodes{{0, Sig{}}};
   std::map<std::pair<int, Sig>, int> node_id{{nodes[0], 0}};
   auto intern = [&](int rem, Sig sg) {
      auto key = std::make_pair(rem, std::move(sg));
      auto it = node_id.find(key);
      if (it != node_id.end()) return it->second;
      nodes.push_back(key);
      node_id.emplace(std::move(key), static_cast<int>(nodes.size()) - 1);
      return static_cast<int>(nodes.size()) - 1;
   };
   std::vector<std::pair<int, int>> keys;   // (q, node); q = -1: INVALID
   std::map<std::pair<int, int>, int> ids;
   auto state_of = [&](int q, int node) {
      if (q == 0) node = 0;   // dead is dead
      auto key = std::make_pair(q, node);
      auto it = ids.find(key);
      if (it != ids.end()) return it->second;
      keys.push_back(key);
      ids.emplace(key, static_cast<int>(keys.size()) - 1);
      return static_cast<int>(keys.size()) - 1;
   };
   state_of(0, 0);   // state 0 = dead
   const int init = state_of(A.init, 0);
   const int inv = invalid_sink ? state_of(-1, 0) : 0;
   // Only CANONICAL sequences are modelled (shortest form, <= U+10FFFF): the reference decodes overlong and out-of-range forms
   // arithmetically too, but they are as good as absent from real text and cost states; like structure errors they lead to
   // INVALID and the decode pass answers the row exactly.
   const Sig canon[3] = {sig_restrict(full, 0x80u, 0x800u), sig_restrict(full, 0x800u, 0x10000u), sig_restrict(full, 0x10000u, 0x110000u)};
   // What a byte >= 0x80 does to the decoder part of a state depends on the node alone, not on q: worked out once per node, when the
   // first state that carries it is expanded (bytes in ascending order, as the states' own loops run: nodes are numbered in the same
   // order as if every state did the work itself).  kind 0: INVALID; 1: on to node `val`; 2: the character is complete, class `val`.
   struct NodeStep {
      int kind, val;
   };
   std::vector<std::array<NodeStep, 128>> node_steps;   // [node][b - 0x80]
   std::vector<uint8_t> node_done;
   auto steps_of = [&](int nd) {
      if (static_cast<size_t>(nd) >= node_done.size()) {
         node_done.resize(static_cast<size_t>(nd) + 1, 0);
         node_steps.resize(static_cast<size_t>(nd) + 1);
      }
      if (node_done[static_cast<size_t>(nd)]) return;
      std::array<NodeStep, 128> row;
      for (int b = 0x80; b < 256; ++b) {
         NodeStep st{0, 0};
         if (nd == 0) {
            if (b >= 0xC0 && b < 0xF8) {
               const int rem = b < 0xE0 ? 1 : (b < 0xF0 ? 2 : 3);
               const uint32_t pay = static_cast<uint32_t>(b) & (b < 0xE0 ? 0x1Fu : (b < 0xF0 ? 0x0Fu : 0x07u));
               const uint32_t span = 1u << (6 * rem);
               Sig sub = sig_slice(canon[rem - 1], pay * span, (pay + 1) * span);
               if (!sig_all_invalid(sub)) st = NodeStep{1, intern(rem, std::move(sub))};   // (C0, C1, F5..F7: nothing canonical starts here)
            }
         } else if (b < 0xC0) {
            const int rem = nodes[static_cast<size_t>(nd)].first;
            const uint32_t v = static_cast<uint32_t>(b) & 0x3Fu, span = 1u << (6 * (rem - 1));
            Sig sub = sig_slice(nodes[static_cast<size_t>(nd)].second, v * span, (v + 1) * span);
            if (!sig_all_invalid(sub)) {   // (else: overlong form / beyond U+10FFFF)
               if (rem == 1) st = NodeStep{2, sub[0].second};
               else st = NodeStep{1, intern(rem - 1, std::move(sub))};
            }
         }
         row[static_cast<size_t>(b - 0x80)] = st;
      }
      if (static_cast<size_t>(nd) >= node_done.size()) {   // (intern may have added nodes; the arrays are indexed by node)
         node_done.resize(static_cast<size_t>(nd) + 1, 0);
         node_steps.resize(static_cast<size_t>(nd) + 1);
      }
      node_steps[static_cast<size_t>(nd)] = row;
      node_done[static_cast<size_t>(nd)] = 1;
   };
   int ascii_cls[128];
   for (int b = 0; b < 128; ++b) ascii_cls[b] = sig_eval(full, static_cast<uint32_t>(b));
   std::vector<int> T;
   for (size_t s = 0; s < keys.size(); ++s) {
      if (static_cast<int>(keys.size()) > kMaxByteStates) return r;
      const int q = keys[s].first, nd = keys[s].second;
      T.resize((s + 1) * 256);
      if (q > 0) steps_of(nd);
      for (int b = 0; b < 256; ++b) {
         int dst;
         if (q == 0) dst = 0;
         else if (q < 0) dst = static_cast<int>(s);
         else if (b < 0x80) dst = nd == 0 ? state_of(A.T(q, ascii_cls[b]), 0) : inv;
         else {
            const NodeStep st = node_steps[static_cast<size_t>(nd)][static_cast<size_t>(b - 0x80)];
            dst = st.kind == 0 ? inv : (st.kind == 2 ? state_of(A.T(q, st.val), 0) : state_of(q, st.val));
         }
         T[s * 256 + static_cast<size_t>(b)] = dst;
      }
   }
   const int n = static_cast<int>(keys.size());
   r.d.n = n;
   r.d.ncol = 256;
   r.d.T = std::move(T);
   r.d.init = init;
   r.d.out.assign(static_cast<size_t>(n), 0);
   std::vector<int> fin(static_cast<size_t>(n), 0), labels(static_cast<size_t>(n), 0);
   for (int s = 0; s < n; ++s) {
      const int q = keys[static_cast<size_t>(s)].first, nd = keys[static_cast<size_t>(s)].second;
      if (q > 0 && nd == 0) {
         r.d.out[static_cast<size_t>(s)] = (*A.out)[static_cast<size_t>(q)];
         fin[static_cast<size_t>(s)] = fin_class[static_cast<size_t>(q)];
      } else if (q != 0 && invalid_sink) {
         fin[static_cast<size_t>(s)] = 2;   // (`.match.` only: a search never asks where its forward walk stopped, and without the
      }                                     //  label a continuation node whose every path dies merges with the dead state)
      labels[static_cast<size_t>(s)] = fin[static_cast<size_t>(s)] + (q < 0 ? 4 : 0);
   }
   std::vector<int> o2n;
   minimise(r.d, 0, &labels, &o2n);
   r.fin.assign(static_cast<size_t>(r.d.n), 0);
   for (int s = 0; s < n; ++s)
      if (o2n[static_cast<size_t>(s)] >= 0) r.fin[static_cast<size_t>(o2n[static_cast<size_t>(s)])] = fin[static_cast<size_t>(s)];
   r.inv = invalid_sink ? o2n[static_cast<size_t>(inv)] : -1;
   r.ok = true;
   return r;
}

// reverse (the row is scanned right to left: continuation bytes arrive before their lead byte): (r, node); node 0 = between
// characters, else (k continuation bytes seen, class of the character as a function of the payload bits still to come)
ByteDfa build_reverse_bytes(const ClassDfaView& R, const Sig& full) {
   ByteDfa r;
   // node: k continuation bytes seen, and for every length n = k+1 .. 4 the character could still have, its class as a function
   // of the payload bits still to come -- derived from the class function restricted to what is CANONICAL at that length (see
   // build_forward_bytes), so an overlong or out-of-range form evaluates to -1 when its lead byte arrives
   using Node = std::pair<int, std::array<Sig, 3>>;
   std::vector<Node> nodes{Node{0, {}}};
   std::map<Node, int> node_id{{nodes[0], 0}};
   auto intern = [&](int k, std::array<Sig, 3> sg) {
      Node key{k, std::move(sg)};
      auto it = node_id.find(key);
      if (it != node_id.end()) return it->second;
      nodes.push_back(key);
      node_id.emplace(std::move(key), static_cast<int>(nodes.size()) - 1);
      return static_cast<int>(nodes.size()) - 1;
   };
   std::vector<std::pair<int, int>> keys;   // (r, node); r = -1: INVALID
   std::map<std::pair<int, int>, int> ids;
   auto state_of = [&](int q, int node) {
      auto key = std::make_pair(q, node);
      auto it = ids.find(key);
      if (it != ids.end()) return it->second;
      keys.push_back(key);
      ids.emplace(key, static_cast<int>(keys.size()) - 1);
      return static_cast<int>(keys.size()) - 1;
   };
   const int init = state_of(R.init, 0);
   const int inv = state_of(-1, 0);
   const std::array<Sig, 3> canon = {sig_restrict(full, 0x80u, 0x800u), sig_restrict(full, 0x800u, 0x10000u), sig_restrict(full, 0x10000u, 0x110000u)};
   // k continuation bytes seen -> k + 1: the lengths still possible are n >= k + 2 (index i = n - 2 >= k); each takes the payload
   auto shrink_all = [&](const std::array<Sig, 3>& from, int k, uint32_t v) {
      std::array<Sig, 3> o;
      bool any = false;
      for (int i = k; i < 3; ++i) {
         o[static_cast<size_t>(i)] = sig_shrink(from[static_cast<size_t>(i)], v, 1u << (21 - 6 * (k + 1)));
         if (!sig_all_invalid(o[static_cast<size_t>(i)])) any = true;
      }
      return std::make_pair(o, any);
   };
   // The node a continuation byte leads to depends on the node alone, not on q: worked out once per node, when the first state that
   // carries it is expanded (payloads in ascending order, as the states' own loops run: same node numbering).  -1 = INVALID.
   std::vector<std::array<int, 64>> cont_next;   // [node][b & 0x3F]
   std::vector<uint8_t> cont_done;
   auto cont_of = [&](int nd) {
      if (static_cast<size_t>(nd) >= cont_done.size()) {
         cont_done.resize(static_cast<size_t>(nd) + 1, 0);
         cont_next.resize(static_cast<size_t>(nd) + 1);
      }
      if (cont_done[static_cast<size_t>(nd)]) return;
      std::array<int, 64> row;
      const int k = nd == 0 ? 0 : nodes[static_cast<size_t>(nd)].first;
      for (uint32_t v = 0; v < 64u; ++v) {
         if (k == 3) {
            row[v] = -1;
            continue;
         }
         auto pr = nd == 0 ? shrink_all(canon, 0, v) : shrink_all(nodes[static_cast<size_t>(nd)].second, k, v);
         row[v] = pr.second ? intern(k + 1, std::move(pr.first)) : -1;
      }
      if (static_cast<size_t>(nd) >= cont_done.size()) {
         cont_done.resize(static_cast<size_t>(nd) + 1, 0);
         cont_next.resize(static_cast<size_t>(nd) + 1);
      }
      cont_next[static_cast<size_t>(nd)] = row;
      cont_done[static_cast<size_t>(nd)] = 1;
   };
   int ascii_cls[128];
   for (int b = 0; b < 128; ++b) ascii_cls[b] = sig_eval(full, static_cast<uint32_t>(b));
   std::vector<int> T;
   for (size_t s = 0; s < keys.size(); ++s) {
      if (static_cast<int>(keys.size()) > kMaxByteStates) return r;
      const int q = keys[s].first, nd = keys[s].second;
      T.resize((s + 1) * 256);
      if (q >= 0) cont_of(nd);
      for (int b = 0; b < 256; ++b) {
         int dst;
         const bool cont = b >= 0x80 && b < 0xC0, lead = b >= 0xC0 && b < 0xF8;
         if (q < 0) dst = static_cast<int>(s);
         else if (nd == 0) {
            if (b < 0x80) dst = state_of(R.T(q, ascii_cls[b]), 0);
            else if (cont) {
               const int nx = cont_next[0][static_cast<size_t>(b) & 0x3Fu];
               dst = nx >= 0 ? state_of(q, nx) : inv;
            } else dst = inv;   // a lead byte with nothing behind it, F8..FF
         } else {
            const int k = nodes[static_cast<size_t>(nd)].first;
            if (cont) {
               const int nx = cont_next[static_cast<size_t>(nd)][static_cast<size_t>(b) & 0x3Fu];
               dst = nx >= 0 ? state_of(q, nx) : inv;
            } else if (lead) {
               const int len = b < 0xE0 ? 2 : (b < 0xF0 ? 3 : 4);
               const uint32_t pay = static_cast<uint32_t>(b) & (b < 0xE0 ? 0x1Fu : (b < 0xF0 ? 0x0Fu : 0x07u));
               if (len - 1 != k) dst = inv;
               else {
                  const int c = sig_eval(nodes[static_cast<size_t>(nd)].second[static_cast<size_t>(len - 2)], pay);
                  dst = c < 0 ? inv : state_of(R.T(q, c), 0);
               }
            } else dst = inv;   // continuation bytes without their lead
         }
         T[s * 256 + static_cast<size_t>(b)] = dst;
      }
   }
   const int n = static_cast<int>(keys.size());
   r.d.n = n;
   r.d.ncol = 256;
   r.d.T = std::move(T);
   r.d.init = init;
   r.d.out.assign(static_cast<size_t>(n), 0);
   std::vector<int> labels(static_cast<size_t>(n), 0);
   for (int s = 0; s < n; ++s) {
      const int q = keys[static_cast<size_t>(s)].first, nd = keys[static_cast<size_t>(s)].second;
      if (q >= 0 && nd == 0) r.d.out[static_cast<size_t>(s)] = (*R.out)[static_cast<size_t>(q)];
      labels[static_cast<size_t>(s)] = q < 0 ? 1 : 0;
   }
   std::vector<int> o2n;
   minimise(r.d, -1, &labels, &o2n);
   r.fin.assign(static_cast<size_t>(r.d.n), 0);
   r.inv = o2n[static_cast<size_t>(inv)];
   r.ok = true;
   return r;
}

struct Blob {
   std::vector<uint8_t> b;
   uint32_t put(const void* p, size_t n) {
      while (b.size() % 16) b.push_back(0);
      uint32_t off = static_cast<uint32_t>(b.size());
      const uint8_t* q = static_cast<const uint8_t*>(p);
      if (n) b.insert(b.end(), q, q + n);
      if (n == 0) b.push_back(0);
      return off;
   }
};

}   // namespace

// FNV-1a over the whole image with the checksum field read as zero
uint32_t blob_checksum(const uint8_t* b, size_t n) {
   const size_t c0 = offsetof(FxpHeader, checksum);
   uint32_t x = 2166136261u;
   for (size_t i = 0; i < n; ++i) {
      const uint8_t v = (i >= c0 && i < c0 + 4) ? 0 : b[i];
      x = (x ^ v) * 16777619u;
   }
   return x;
}

namespace {
Program finish(FxpHeader h, Blob& bl) {
   while (bl.b.size() % 16) bl.b.push_back(0);
   h.magic = FXP_MAGIC;
   h.version = FXP_VERSION;
   h.total_bytes = static_cast<uint32_t>(bl.b.size());
   h.checksum = 0;
   std::memcpy(bl.b.data(), &h, sizeof(h));
   h.checksum = blob_checksum(bl.b.data(), bl.b.size());
   std::memcpy(bl.b.data(), &h, sizeof(h));
   Program p;
   p.status = static_cast<int>(h.status);
   p.blob.swap(bl.b);
   return p;
}

Program invalid_program(int status) {
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = FXP_MODE_INVALID;
   h.status = static_cast<uint32_t>(status);
   return finish(h, bl);
}

std::vector<int32_t> decode_chars(const std::string& s) {
   std::vector<int32_t> out;
   int i = 1, n = static_cast<int>(s.size());
   while (i <= n) {
      int nxt;
      bool valid;
      next_idxutf8_strict(s, i, nxt, valid);
      out.push_back(valid ? ichar_utf8(s.substr(static_cast<size_t>(i - 1), static_cast<size_t>(nxt - i))) : 65535);
      i = nxt;
   }
   return out;
}

bool border_free(const std::string& p) {
   for (size_t k = 1; k < p.size(); ++k)
      if (p.compare(0, k, p, p.size() - k, k) == 0) return false;
   return true;
}

// interval starts, class of each interval, ASCII class table and the two-level BMP page map
void emit_class_map(Blob& bl, FxpHeader& h, const std::vector<int32_t>& bounds, int nI, const std::vector<int>& cls_of) {
   auto interval_of = [&](int32_t code) {
      return static_cast<int>(std::upper_bound(bounds.begin(), bounds.end(), code) - bounds.begin()) - 1;
   };
   h.n_bounds = static_cast<uint32_t>(nI);
   std::vector<int32_t> b32(bounds.begin(), bounds.begin() + nI);
   h.off_bounds = bl.put(b32.data(), b32.size() * 4);
   std::vector<uint16_t> bc(static_cast<size_t>(nI));
   for (int k = 0; k < nI; ++k) bc[static_cast<size_t>(k)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(k)]);
   h.off_bound_cls = bl.put(bc.data(), bc.size() * 2);
   std::vector<uint16_t> ac(128);
   for (int c = 0; c < 128; ++c) ac[static_cast<size_t>(c)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(interval_of(c))]);
   h.off_ascii_cls = bl.put(ac.data(), ac.size() * 2);
   h.cls_nul = static_cast<uint32_t>(cls_of[static_cast<size_t>(interval_of(0))]);
   h.cls_ffff = static_cast<uint32_t>(cls_of[static_cast<size_t>(interval_of(65535))]);
   // two-level class map of the BMP: the on-device decoder classifies a 2-/3-byte character with two table reads
   // instead of a binary search over the interval starts
   std::vector<uint16_t> page_of(1024);
   std::vector<uint16_t> pages;
   std::map<std::vector<uint16_t>, uint16_t> seen;   // page contents -> page number, numbered in order of first appearance
   std::vector<int> uniform_page;                    // class -> number of the page that holds that class 64 times (-1: none yet)
   std::vector<uint16_t> v(64);
   int iv = interval_of(0);                          // interval of the code the walk stands on (the pages are walked in order)
   for (int pg = 0; pg < 1024; ++pg) {
      while (iv + 1 < nI && bounds[static_cast<size_t>(iv) + 1] <= pg * 64) ++iv;
      // most pages lie inside ONE interval: looked up by class, not by contents
      const bool one_interval = iv + 1 >= nI || bounds[static_cast<size_t>(iv) + 1] > pg * 64 + 63;
      const int c0 = cls_of[static_cast<size_t>(iv)];
      if (one_interval && static_cast<size_t>(c0) < uniform_page.size() && uniform_page[static_cast<size_t>(c0)] >= 0) {
         page_of[static_cast<size_t>(pg)] = static_cast<uint16_t>(uniform_page[static_cast<size_t>(c0)]);
         continue;
      }
      bool same = true;
      for (int k = 0; k < 64; ++k) {
         int32_t code = pg * 64 + k;
         while (iv + 1 < nI && bounds[static_cast<size_t>(iv) + 1] <= code) ++iv;
         v[static_cast<size_t>(k)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(iv)]);
         same = same && v[static_cast<size_t>(k)] == v[0];
      }
      auto it = seen.find(v);
      if (it == seen.end()) {
         it = seen.emplace(v, static_cast<uint16_t>(seen.size())).first;
         pages.insert(pages.end(), v.begin(), v.end());
      }
      page_of[static_cast<size_t>(pg)] = it->second;
      if (same) {
         if (uniform_page.size() <= static_cast<size_t>(v[0])) uniform_page.resize(static_cast<size_t>(v[0]) + 1, -1);
         uniform_page[static_cast<size_t>(v[0])] = it->second;
      }
   }
   h.n_pages = static_cast<uint32_t>(seen.size());
   h.off_cls_page = bl.put(page_of.data(), page_of.size() * 2);
   h.off_cls_pages = bl.put(pages.data(), pages.size() * 2);
}

std::vector<uint32_t> to_words(const Bits& b, uint32_t words) {
   std::vector<uint32_t> w(words, 0);
   for (uint32_t i = 0; i < words; ++i) {
      const uint64_t q = (i >> 1) < b.size() ? b[i >> 1] : 0;
      w[i] = static_cast<uint32_t>(q >> ((i & 1) * 32));
   }
   return w;
}

// The DFA does not fit: emit the NFA itself for on-device simulation of state SETS (same algorithm as the DFA path --
// reverse unanchored pass for the leftmost start, forward anchored pass for the longest end -- with bitsets as states).
Program compile_nfa_sim(const Nfa& nfa, const Literals& lit, int op, const std::vector<int32_t>& bounds, int nI,
                        const std::vector<FlatTra>& tras, const std::vector<Bits>& clos, const std::vector<Bits>& rclos) {
   const int N = nfa.nfa_top;
   const uint32_t words = static_cast<uint32_t>(N + 32) / 32;
   // classes: intervals consumed by exactly the same transitions
   std::vector<std::vector<int>> sig(static_cast<size_t>(nI));
   for (size_t t = 0; t < tras.size(); ++t)
      for (int k : tras[t].acc) sig[static_cast<size_t>(k)].push_back(static_cast<int>(t));
   std::vector<int> cls_of(static_cast<size_t>(nI));
   std::map<std::vector<int>, int> m;
   for (int k = 0; k < nI; ++k) {
      auto it = m.find(sig[static_cast<size_t>(k)]);
      if (it == m.end()) it = m.emplace(sig[static_cast<size_t>(k)], static_cast<int>(m.size())).first;
      cls_of[static_cast<size_t>(k)] = it->second;
   }
   const int ncls = static_cast<int>(m.size());
   const size_t table_words = static_cast<size_t>(ncls) * (N + 1) * words;
   if (table_words * 8 > (size_t(1) << 30)) return invalid_program(FX_ERR_NFA_LIMIT);   // 2 tables x 4 bytes per word: keep under 1 GiB

   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = op == OP_SEARCH ? FXP_MODE_SEARCH_ENGINE : FXP_MODE_MATCH_ENGINE;
   h.flags = FXP_F_NFA_SIM | FXP_F_HAS_R;
   h.n_classes = static_cast<uint32_t>(ncls);
   h.len_prefix = static_cast<uint32_t>(lit.prefix.size());
   h.len_suffix = static_cast<uint32_t>(lit.suffix.size());
   h.len_all = static_cast<uint32_t>(lit.all.size());
   if (bt(clos[static_cast<size_t>(nfa.entry)], nfa.exit)) h.flags |= FXP_F_INIT_ACCEPTING;
   if (!f_eq(lit.prefix, "")) h.flags |= FXP_F_PREFILTER;
   if (!f_eq(lit.suffix, "")) h.flags |= FXP_F_HAS_SUFFIX;
   if (op == OP_MATCH && !f_eq(lit.all, "")) h.flags |= FXP_F_MATCH_LITERAL;
   emit_class_map(bl, h, bounds, nI, cls_of);
   h.off_prefix = bl.put(lit.prefix.data(), lit.prefix.size());
   h.off_suffix = bl.put(lit.suffix.data(), lit.suffix.size());
   h.off_all = bl.put(lit.all.data(), lit.all.size());
   h.nfa_N = static_cast<uint32_t>(N);
   h.nfa_words = words;
   h.nfa_entry = static_cast<uint32_t>(nfa.entry);
   h.nfa_exit = static_cast<uint32_t>(nfa.exit);
   // per-class transition bitsets
   std::vector<uint32_t> fwdT(table_words, 0), revT(table_words, 0);
   std::vector<std::vector<uint8_t>> acc_cls(tras.size(), std::vector<uint8_t>(static_cast<size_t>(ncls), 0));
   for (size_t t = 0; t < tras.size(); ++t)
      for (int k : tras[t].acc) acc_cls[t][static_cast<size_t>(cls_of[static_cast<size_t>(k)])] = 1;
   for (size_t t = 0; t < tras.size(); ++t) {
      const std::vector<uint32_t> cw = to_words(clos[static_cast<size_t>(tras[t].dst)], words);
      const std::vector<uint32_t> rw = to_words(rclos[static_cast<size_t>(tras[t].src)], words);
      for (int c = 0; c < ncls; ++c) {
         if (!acc_cls[t][static_cast<size_t>(c)]) continue;
         uint32_t* f = &fwdT[(static_cast<size_t>(c) * (N + 1) + tras[t].src) * words];
         uint32_t* r = &revT[(static_cast<size_t>(c) * (N + 1) + tras[t].dst) * words];
         for (uint32_t i = 0; i < words; ++i) {
            f[i] |= cw[i];
            r[i] |= rw[i];
         }
      }
   }
   const std::vector<uint32_t> init = to_words(clos[static_cast<size_t>(nfa.entry)], words);
   const std::vector<uint32_t> f0 = to_words(rclos[static_cast<size_t>(nfa.exit)], words);
   std::vector<uint32_t> rstart = f0;   // F0 + pre(NUL, F0), hit cleared: the trailing NUL is never a start
   for (int z = 1; z <= N; ++z)
      if ((f0[static_cast<size_t>(z) >> 5] >> (z & 31)) & 1u) {
         const uint32_t* r = &revT[(static_cast<size_t>(h.cls_nul) * (N + 1) + z) * words];
         for (uint32_t i = 0; i < words; ++i) rstart[i] |= r[i];
      }
   h.off_nfa_init = bl.put(init.data(), init.size() * 4);
   h.off_nfa_f0 = bl.put(f0.data(), f0.size() * 4);
   h.off_nfa_rstart = bl.put(rstart.data(), rstart.size() * 4);
   h.off_nfa_fwd = bl.put(fwdT.data(), fwdT.size() * 4);
   h.off_nfa_rev = bl.put(revT.data(), revT.size() * 4);
   // empty placeholders so that every offset stays inside the blob
   const uint32_t none = bl.put(nullptr, 0);
   h.off_TA = h.off_TR = h.off_accA = h.off_hitR = h.off_finalM = h.off_fastA = h.off_fastR = none;
   h.off_chain_cls = h.off_chain_TR = h.off_chain_TA = none;
   return finish(h, bl);
}

}   // namespace

Program compile_from_nfa(const Nfa& nfa, const Literals& lit, int op, const Limits& lim) {
   if (nfa.status != SYNTAX_VALID) return invalid_program(nfa.status);
   const int N = nfa.nfa_top;
   if (N > lim.max_nfa_states) return invalid_program(FX_ERR_NFA_LIMIT);
   const size_t W = (static_cast<size_t>(N) + 64) / 64;

   // ---- 1. code-point intervals induced by every segment edge ---------------------------------------
   std::vector<int32_t> bounds{0, 0x200000};
   for (int i = 1; i <= N; ++i)
      for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward)
         for (const Seg& s : tr.c) {
            if (s.max < 0 || s.min > s.max || s.min >= 0x200000) continue;
            bounds.push_back(std::max<int32_t>(s.min, 0));
            bounds.push_back(std::min<int32_t>(s.max, 0x1FFFFF) + 1);
         }
   std::sort(bounds.begin(), bounds.end());
   bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
   const int nI = static_cast<int>(bounds.size()) - 1;
   auto interval_of = [&](int32_t code) {
      return static_cast<int>(std::upper_bound(bounds.begin(), bounds.end(), code) - bounds.begin()) - 1;
   };

   // ---- 2. flatten transitions, epsilon closures -----------------------------------------------------
   std::vector<FlatTra> tras;
   std::vector<std::vector<int>> fwd(static_cast<size_t>(N) + 1), inc(static_cast<size_t>(N) + 1), eps(static_cast<size_t>(N) + 1);
   for (int i = 1; i <= N; ++i)
      for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward) {
         if (tr.dst == NFA_NULL_TRANSITION || tr.dst < 1 || tr.dst > N) continue;
         if (tr.is_epsilon()) eps[static_cast<size_t>(i)].push_back(tr.dst);
         FlatTra ft;
         ft.src = i;
         ft.dst = tr.dst;
         for (int k = 0; k < nI; ++k)
            if (tr.accepts(bounds[static_cast<size_t>(k)])) ft.acc.push_back(k);
         if (!ft.acc.empty()) {
            fwd[static_cast<size_t>(i)].push_back(static_cast<int>(tras.size()));
            inc[static_cast<size_t>(tr.dst)].push_back(static_cast<int>(tras.size()));
            tras.push_back(std::move(ft));
         }
      }
   std::vector<Bits> clos(static_cast<size_t>(N) + 1, Bits(W, 0)), rclos(static_cast<size_t>(N) + 1, Bits(W, 0));
   for (int z = 1; z <= N; ++z) {
      std::vector<int> st{z};
      bs(clos[static_cast<size_t>(z)], z);
      while (!st.empty()) {
         int x = st.back();
         st.pop_back();
         for (int y : eps[static_cast<size_t>(x)])
            if (!bt(clos[static_cast<size_t>(z)], y)) {
               bs(clos[static_cast<size_t>(z)], y);
               st.push_back(y);
            }
      }
      for (int x = 1; x <= N; ++x)
         if (bt(clos[static_cast<size_t>(z)], x)) bs(rclos[static_cast<size_t>(x)], z);
   }

   // ---- 3. forward anchored DFA A ----------------------------------------------------------------------
   Dfa A;
   A.ncol = nI;
   {
      std::map<Bits, int> ids;
      std::vector<Bits> sets;
      sets.emplace_back(W, 0);   // state 0 = dead = empty set (DFA_INVALID_INDEX)
      ids.emplace(sets[0], 0);
      sets.push_back(clos[static_cast<size_t>(nfa.entry)]);
      ids.emplace(sets[1], 1);
      A.init = 1;
      std::vector<Bits> next(static_cast<size_t>(nI), Bits(W, 0));
      for (size_t s = 0; s < sets.size(); ++s) {
         for (auto& nb : next) std::fill(nb.begin(), nb.end(), 0);
         const Bits cur = sets[s];
         for (int x = 1; x <= N; ++x) {
            if (!bt(cur, x)) continue;
            for (int ti : fwd[static_cast<size_t>(x)]) {
               const FlatTra& ft = tras[static_cast<size_t>(ti)];
               for (int k : ft.acc) bor(next[static_cast<size_t>(k)], clos[static_cast<size_t>(ft.dst)]);
            }
         }
         A.T.resize((s + 1) * static_cast<size_t>(nI));
         for (int k = 0; k < nI; ++k) {
            auto it = ids.find(next[static_cast<size_t>(k)]);
            if (it == ids.end()) {
               if (static_cast<int>(sets.size()) >= lim.max_dfa_states) return compile_nfa_sim(nfa, lit, op, bounds, nI, tras, clos, rclos);
               it = ids.emplace(next[static_cast<size_t>(k)], static_cast<int>(sets.size())).first;
               sets.push_back(next[static_cast<size_t>(k)]);
            }
            A.T[s * static_cast<size_t>(nI) + static_cast<size_t>(k)] = it->second;
         }
      }
      A.n = static_cast<int>(sets.size());
      A.out.assign(static_cast<size_t>(A.n), 0);
      for (int s = 0; s < A.n; ++s) A.out[static_cast<size_t>(s)] = bt(sets[static_cast<size_t>(s)], nfa.exit) ? 1 : 0;
   }
   minimise(A, 0);

   // ---- 4. reverse unanchored DFA R (search only) ---------------------------------------------------------
   Dfa R;
   R.ncol = nI;
   R.ok = false;
   bool r_has_skip = false;   // R distinguishes the SKIP symbol (fast path may translate UTF-8 in place)
   int R_start_raw = 0;
   const int i_nul = interval_of(0);
   if (op == OP_SEARCH) {
      R.ok = true;
      const Bits& F0 = rclos[static_cast<size_t>(nfa.exit)];
      std::map<std::pair<Bits, int>, int> ids;
      std::vector<std::pair<Bits, int>> sets;
      sets.emplace_back(F0, 0);
      ids.emplace(sets[0], 0);
      R.init = 0;
      std::vector<Bits> ac(static_cast<size_t>(nI), Bits(W, 0));
      for (size_t s = 0; s < sets.size() && R.ok; ++s) {
         for (auto& b : ac) std::fill(b.begin(), b.end(), 0);
         const Bits cur = sets[s].first;
         for (int z = 1; z <= N; ++z) {
            if (!bt(cur, z)) continue;
            for (int ti : inc[static_cast<size_t>(z)]) {
               const FlatTra& ft = tras[static_cast<size_t>(ti)];
               for (int k : ft.acc) bs(ac[static_cast<size_t>(k)], ft.src);
            }
         }
         R.T.resize((s + 1) * static_cast<size_t>(nI));
         for (int k = 0; k < nI; ++k) {
            Bits pre(W, 0);
            for (int x = 1; x <= N; ++x)
               if (bt(ac[static_cast<size_t>(k)], x)) bor(pre, rclos[static_cast<size_t>(x)]);
            int hit = bt(pre, nfa.entry) ? 1 : 0;
            bor(pre, F0);
            std::pair<Bits, int> key(pre, hit);
            auto it = ids.find(key);
            if (it == ids.end()) {
               if (static_cast<int>(sets.size()) >= lim.max_dfa_states) {
                  R.ok = false;
                  break;
               }
               it = ids.emplace(key, static_cast<int>(sets.size())).first;
               sets.push_back(key);
            }
            R.T[s * static_cast<size_t>(nI) + static_cast<size_t>(k)] = it->second;
         }
      }
      if (R.ok) {
         R.n = static_cast<int>(sets.size());
         R.out.assign(static_cast<size_t>(R.n), 0);
         for (int s = 0; s < R.n; ++s) R.out[static_cast<size_t>(s)] = static_cast<uint8_t>(sets[static_cast<size_t>(s)].second);
         // start of the scan = state after the trailing NUL with its hit cleared (the trailing NUL is never a start:
         // `do while (start < len(str))`, api_internal_m.F90:108).  Add that state explicitly before minimising.
         int t = R.T[static_cast<size_t>(R.init) * nI + static_cast<size_t>(i_nul)];
         std::pair<Bits, int> key(sets[static_cast<size_t>(t)].first, 0);
         auto it = ids.find(key);
         if (it != ids.end()) {
            R_start_raw = it->second;
         } else {
            // same W, hit = 0: transitions identical to state t (they depend on W only)
            R_start_raw = R.n;
            R.n += 1;
            R.out.push_back(0);
            R.T.resize(static_cast<size_t>(R.n) * nI);
            for (int k = 0; k < nI; ++k) R.T[static_cast<size_t>(R_start_raw) * nI + k] = R.T[static_cast<size_t>(t) * nI + k];
         }
         R.init = R_start_raw;
         // Extra column SKIP (index nI): the symbol the fast kernel substitutes for continuation bytes inside a valid
         // multi-byte character.  It must leave W unchanged and CLEAR the hit (a hit belongs to the character's first
         // byte only), i.e. (W,1) -> (W,0), (W,0) -> itself.  The twins (W,0) are added where missing.
         const int nC = nI + 1;
         {
            std::vector<int> T2(static_cast<size_t>(R.n) * nC);
            for (int st = 0; st < R.n; ++st)
               for (int k = 0; k < nI; ++k) T2[static_cast<size_t>(st) * nC + k] = R.T[static_cast<size_t>(st) * nI + k];
            std::map<std::vector<int>, int> twin_of_row;   // states with hit = 0, keyed by their transition row (same W <=> same row
                                                           // is not guaranteed, so twins are created per hit state and merged by minimise)
            int n0 = R.n;
            for (int st = 0; st < n0; ++st) {
               if (!R.out[static_cast<size_t>(st)]) {
                  T2[static_cast<size_t>(st) * nC + nI] = st;
                  continue;
               }
               int tw = R.n++;
               R.out.push_back(0);
               T2.resize(static_cast<size_t>(R.n) * nC);
               for (int k = 0; k < nI; ++k) T2[static_cast<size_t>(tw) * nC + k] = T2[static_cast<size_t>(st) * nC + k];
               T2[static_cast<size_t>(tw) * nC + nI] = tw;
               T2[static_cast<size_t>(st) * nC + nI] = tw;
            }
            R.T.swap(T2);
            R.ncol = nC;
         }
         Dfa Ru = R;
         minimise(Ru, -1);
         if (Ru.n <= 8) {
            R = Ru;
            r_has_skip = true;
         } else {
            // more than 8 states once SKIP must be told apart: if the SKIP-blind automaton still fits the v_perm tables keep
            // that ASCII-only fast path, otherwise keep SKIP (class-indexed chain tables have no 8-state limit)
            Dfa Rn = R;
            for (int st = 0; st < Rn.n; ++st) Rn.T[static_cast<size_t>(st) * nC + nI] = st;
            minimise(Rn, -1);
            if (Rn.n <= 8) {
               R = Rn;
            } else {
               R = Ru;
               r_has_skip = true;
            }
         }
      }
   }

   // (used by 4b and by section 7, where the argument is written out)
   auto suffix_is_necessary_ending = [&]() -> bool {
      bool ok = lit.suffix.find('\0') == std::string::npos;
      // (Literals with NON-ASCII characters qualify since round 4.  The argument is made on characters and carries over to the bytes
      //  the driver's INDEX works on.  The property used: every match is LONGER than the suffix (the length check below), so the suffix
      //  occurrence that ends a match lies at least one character -- at least one byte -- behind the match's start; prefix and suffix
      //  MAY overlap inside a match (`ab{2,}`: prefix `abb`, suffix `bb`).  On pure-ASCII rows a non-ASCII literal never occurs: a
      //  necessary one means "no match", which is what the driver
      //  (prefix absent: brute force; suffix absent: no match) and the brute-force scan both say.  On rows the byte-level tables
      //  answer themselves -- valid CANONICAL UTF-8: overlong forms, which decode to the same code point but are other bytes, send a
      //  row of such a program to the general procedure -- byte occurrences are exactly the character-aligned ones.
      //  tests/support/fuzz_prefilter.py with FX_FUZZ_UTF8=1: see DESIGN.md 3.6.)
      const std::vector<int32_t> sc = decode_chars(lit.suffix);
      Bits Wb = rclos[static_cast<size_t>(nfa.exit)];
      for (size_t i = sc.size(); ok && i-- > 0;) {
         if (bt(Wb, nfa.entry)) ok = false;
         const int iv = interval_of(sc[i]);
         if (!(bounds[static_cast<size_t>(iv)] == sc[i] && bounds[static_cast<size_t>(iv) + 1] == sc[i] + 1)) ok = false;
         Bits nxt(W, 0);
         for (int z = 1; ok && z <= N; ++z) {
            if (!bt(Wb, z)) continue;
            for (int ti : inc[static_cast<size_t>(z)]) {
               const FlatTra& ft = tras[static_cast<size_t>(ti)];
               for (int k : ft.acc) {
                  if (k != iv) ok = false;   // some other symbol can stand at this distance from the end of a match
                  else bor(nxt, rclos[static_cast<size_t>(ft.src)]);
               }
            }
         }
         Wb = nxt;
         if (!bany(Wb)) ok = false;
      }
      // shortest accepted string (in symbols) >= prefix + suffix: breadth-first over A
      if (ok) {
         std::vector<int> dist(static_cast<size_t>(A.n), -1);
         std::vector<int> queue{A.init};
         dist[static_cast<size_t>(A.init)] = 0;
         int shortest = -1;
         for (size_t qi = 0; qi < queue.size() && shortest < 0; ++qi) {
            const int st = queue[qi];
            if (A.out[static_cast<size_t>(st)] && st != A.init) {
               shortest = dist[static_cast<size_t>(st)];
               break;
            }
            for (int k = 0; k < nI; ++k) {
               const int t = A.T[static_cast<size_t>(st) * nI + k];
               if (t != 0 && dist[static_cast<size_t>(t)] < 0) {
                  dist[static_cast<size_t>(t)] = dist[static_cast<size_t>(st)] + 1;
                  queue.push_back(t);
               }
            }
         }
         // (round 4: "longer than the suffix" is all the driver's arithmetic needs -- a match that starts at wrapped index st and ends
         //  with the suffix has its suffix occurrence at u = st + m - |suffix| >= st + 1, so the last occurrence in the row lies behind
         //  st and the cut-off `suf_idx(text) < ci(wrapped)`, i.e. ci >= last occurrence (api_internal_m.F90:114-116), never drops st;
         //  rounds 1-3 asked for m >= |prefix| + |suffix|, which sent `ab{2,}` (prefix `abb`, suffix `bb`) to the general kernel.  A match
         //  that IS the suffix -- `A{1,2}bb` on `Abb`: prefix `A`, suffix `Abb` -- is cut off by the reference when no later occurrence
         //  follows: such programs keep the statement-for-statement driver.)
         if (shortest < 0 || static_cast<size_t>(shortest) < sc.size() + 1) ok = false;
      }
      return ok;
   };
   // ---- 4b. prefix literal with a BORDER (`--x`, `aa[bc]`, `abab.*`): the reference's candidate list holds NON-overlapping
   // occurrences of the prefix (utility_m.f90:94-116), so it differs from brute force exactly on rows where two occurrences
   // overlap -- rows that contain a WITNESS p + p[b..] for a border length b.  R is composed with an Aho-Corasick detector of the
   // (reversed) witnesses whose hit is one absorbing state: the tile kernels search by brute force and hand rows that end there
   // to the general engine.  Only with a necessary prefix (checked here on the intervals).  The general engine's own brute-force
   // fallback (api_internal_m.F90:79-81) walks this same R: without a suffix literal it only runs on rows without the prefix, where
   // the detector stays idle; with one it also runs when the first prefix occurrence lies behind the last suffix occurrence, and
   // then -- the suffix being a necessary ending, no match shorter than prefix + suffix -- the row holds no match at all, which is
   // what R reports whether or not it falls into the absorbing state on the way.
   bool overlap_sink = false;
   int R_inv_state = -1;
   if (op == OP_SEARCH && R.ok && !f_eq(lit.prefix, "") && !border_free(lit.prefix) && lit.prefix.find('\0') == std::string::npos &&
       (f_eq(lit.suffix, "") || suffix_is_necessary_ending())) {
      const std::vector<int32_t> pc = decode_chars(lit.prefix);
      const int lp = static_cast<int>(pc.size());
      std::vector<int> piv(static_cast<size_t>(lp));
      bool ok = lp >= 2;
      int q = A.init;
      for (int i = 0; ok && i < lp; ++i) {
         if (i > 0 && A.out[static_cast<size_t>(q)]) ok = false;
         const int iv = interval_of(pc[static_cast<size_t>(i)]);
         piv[static_cast<size_t>(i)] = iv;
         if (!(bounds[static_cast<size_t>(iv)] == pc[static_cast<size_t>(i)] && bounds[static_cast<size_t>(iv) + 1] == pc[static_cast<size_t>(i)] + 1)) ok = false;
         for (int k = 0; ok && k < nI; ++k)
            if (k != iv && A.T[static_cast<size_t>(q) * nI + k] != 0) ok = false;
         if (ok) {
            q = A.T[static_cast<size_t>(q) * nI + iv];
            if (q == 0) ok = false;
         }
      }
      if (ok) {
         // reversed witnesses (R reads right to left)
         std::vector<std::vector<int>> wit;
         for (int b = 1; b < lp; ++b) {
            bool border = true;
            for (int i = 0; border && i < b; ++i) border = pc[static_cast<size_t>(i)] == pc[static_cast<size_t>(lp - b + i)];
            if (!border) continue;
            std::vector<int> w(piv.begin(), piv.end());
            w.insert(w.end(), piv.begin() + b, piv.end());
            std::reverse(w.begin(), w.end());
            wit.push_back(std::move(w));
         }
         // Aho-Corasick states = the prefixes of the witnesses; node 0 = empty
         std::vector<std::vector<int>> node{std::vector<int>{}};
         std::map<std::vector<int>, int> node_id{{node[0], 0}};
         for (const auto& w : wit)
            for (size_t len = 1; len <= w.size(); ++len) {
               std::vector<int> u(w.begin(), w.begin() + static_cast<long>(len));
               if (node_id.emplace(u, static_cast<int>(node.size())).second) node.push_back(u);
            }
         std::vector<uint8_t> terminal(node.size(), 0);
         for (const auto& w : wit) terminal[static_cast<size_t>(node_id[w])] = 1;
         auto ac_step = [&](int nd, int iv) -> int {   // longest suffix of node + iv that is a node; -1 = a witness is complete
            std::vector<int> u = node[static_cast<size_t>(nd)];
            u.push_back(iv);
            for (size_t cut = 0; cut <= u.size(); ++cut) {
               auto it = node_id.find(std::vector<int>(u.begin() + static_cast<long>(cut), u.end()));
               if (it != node_id.end()) {
                  // a witness that ends here (this node or a suffix of it) completes the detection
                  for (size_t c2 = cut; c2 < u.size(); ++c2) {
                     auto it2 = node_id.find(std::vector<int>(u.begin() + static_cast<long>(c2), u.end()));
                     if (it2 != node_id.end() && terminal[static_cast<size_t>(it2->second)]) return -1;
                  }
                  return it->second;
               }
            }
            return 0;
         };
         const int nC = R.ncol;
         std::vector<std::pair<int, int>> keys;   // (r, ac node); (-1, 0) = the sink
         std::map<std::pair<int, int>, int> ids;
         auto state_of = [&](int r, int nd) {
            auto key = std::make_pair(r, nd);
            auto it = ids.find(key);
            if (it != ids.end()) return it->second;
            keys.push_back(key);
            ids.emplace(key, static_cast<int>(keys.size()) - 1);
            return static_cast<int>(keys.size()) - 1;
         };
         const int init = state_of(R.init, 0);
         const int sink = state_of(-1, 0);
         std::vector<int> T2;
         bool fits = true;
         for (size_t s2 = 0; s2 < keys.size() && fits; ++s2) {
            if (keys.size() > 4096) fits = false;
            const int r = keys[s2].first, nd = keys[s2].second;
            T2.resize((s2 + 1) * static_cast<size_t>(nC));
            for (int k = 0; k < nC; ++k) {
               int dst;
               if (r < 0) dst = sink;
               else if (k >= nI) dst = state_of(R.T[static_cast<size_t>(r) * nC + k], nd);   // SKIP column: not a text symbol
               else {
                  const int nn = ac_step(nd, k);
                  dst = nn < 0 ? sink : state_of(R.T[static_cast<size_t>(r) * nC + k], nn);
               }
               T2[s2 * static_cast<size_t>(nC) + static_cast<size_t>(k)] = dst;
            }
         }
         if (fits) {
            Dfa R2;
            R2.n = static_cast<int>(keys.size());
            R2.ncol = nC;
            R2.T = std::move(T2);
            R2.init = init;
            R2.ok = true;
            R2.out.assign(static_cast<size_t>(R2.n), 0);
            std::vector<int> labels(static_cast<size_t>(R2.n), 0);
            for (int s2 = 0; s2 < R2.n; ++s2) {
               if (keys[static_cast<size_t>(s2)].first >= 0) R2.out[static_cast<size_t>(s2)] = R.out[static_cast<size_t>(keys[static_cast<size_t>(s2)].first)];
               else labels[static_cast<size_t>(s2)] = 1;
            }
            std::vector<int> o2n;
            minimise(R2, -1, &labels, &o2n);
            R = R2;
            R_inv_state = o2n[static_cast<size_t>(sink)];
            overlap_sink = true;
         }
      }
   }

   // ---- 5. merge intervals with identical columns into classes -----------------------------------------------
   std::vector<int> cls_of(static_cast<size_t>(nI));
   int ncls = 0;
   {
      std::map<std::vector<int>, int> m;
      for (int k = 0; k < nI; ++k) {
         std::vector<int> col;
         for (int s = 0; s < A.n; ++s) col.push_back(A.T[static_cast<size_t>(s) * nI + k]);
         if (R.ok)
            for (int s = 0; s < R.n; ++s) col.push_back(R.T[static_cast<size_t>(s) * R.ncol + k]);
         auto it = m.find(col);
         if (it == m.end()) it = m.emplace(col, static_cast<int>(m.size())).first;
         cls_of[static_cast<size_t>(k)] = it->second;
      }
      ncls = static_cast<int>(m.size());
   }
   std::vector<int> rep_interval(static_cast<size_t>(ncls), -1);
   for (int k = 0; k < nI; ++k)
      if (rep_interval[static_cast<size_t>(cls_of[static_cast<size_t>(k)])] < 0) rep_interval[static_cast<size_t>(cls_of[static_cast<size_t>(k)])] = k;
   auto TA = [&](int s, int c) { return A.T[static_cast<size_t>(s) * nI + rep_interval[static_cast<size_t>(c)]]; };
   auto TR = [&](int s, int c) { return R.T[static_cast<size_t>(s) * R.ncol + rep_interval[static_cast<size_t>(c)]]; };
   auto class_of_code = [&](int32_t code) { return cls_of[static_cast<size_t>(interval_of(code))]; };

   // ---- 6. emit ---------------------------------------------------------------------------------------------------
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.status = 0;
   h.mode = op == OP_SEARCH ? FXP_MODE_SEARCH_ENGINE : FXP_MODE_MATCH_ENGINE;
   h.n_classes = static_cast<uint32_t>(ncls);
   h.n_bounds = static_cast<uint32_t>(nI);
   h.nA = static_cast<uint32_t>(A.n);
   h.nR = R.ok ? static_cast<uint32_t>(R.n) : 0;
   h.A_init = static_cast<uint32_t>(A.init);
   h.R_start = R.ok ? static_cast<uint32_t>(R.init) : 0;
   h.cls_nul = static_cast<uint32_t>(class_of_code(0));
   h.cls_ffff = static_cast<uint32_t>(class_of_code(65535));
   {
      int t = TA(A.init, static_cast<int>(h.cls_nul));
      h.M_start = static_cast<uint32_t>(t != 0 ? t : A.init);   // api_internal_m.F90:280-289
   }
   h.len_prefix = static_cast<uint32_t>(lit.prefix.size());
   h.len_suffix = static_cast<uint32_t>(lit.suffix.size());
   h.len_all = static_cast<uint32_t>(lit.all.size());
   if (A.out[static_cast<size_t>(A.init)]) h.flags |= FXP_F_INIT_ACCEPTING;
   const bool prefilter = !f_eq(lit.prefix, "");
   const bool has_suffix = !f_eq(lit.suffix, "");
   if (prefilter) h.flags |= FXP_F_PREFILTER;
   if (has_suffix) h.flags |= FXP_F_HAS_SUFFIX;
   if (R.ok) h.flags |= FXP_F_HAS_R;
   if (op == OP_MATCH && !f_eq(lit.all, "")) h.flags |= FXP_F_MATCH_LITERAL;

   emit_class_map(bl, h, bounds, nI, cls_of);
   std::vector<uint16_t> ac(128);
   for (int c = 0; c < 128; ++c) ac[static_cast<size_t>(c)] = static_cast<uint16_t>(class_of_code(c));
   std::vector<uint16_t> ta(static_cast<size_t>(A.n) * ncls);
   for (int s = 0; s < A.n; ++s)
      for (int c = 0; c < ncls; ++c) {
         int t = TA(s, c);
         ta[static_cast<size_t>(s) * ncls + c] = static_cast<uint16_t>(t | (A.out[static_cast<size_t>(t)] ? FXP_FLAG_BIT : 0));
      }
   h.off_TA = bl.put(ta.data(), ta.size() * 2);
   std::vector<uint16_t> tr;
   if (R.ok) {
      tr.resize(static_cast<size_t>(R.n) * ncls);
      for (int s = 0; s < R.n; ++s)
         for (int c = 0; c < ncls; ++c) {
            int t = TR(s, c);
            tr[static_cast<size_t>(s) * ncls + c] = static_cast<uint16_t>(t | (R.out[static_cast<size_t>(t)] ? FXP_FLAG_BIT : 0));
         }
   }
   h.off_TR = bl.put(tr.data(), tr.size() * 2);
   h.off_accA = bl.put(A.out.data(), A.out.size());
   h.off_hitR = bl.put(R.out.data(), R.ok ? R.out.size() : 0);
   std::vector<uint8_t> fin(static_cast<size_t>(A.n));
   for (int s = 0; s < A.n; ++s)   // api_internal_m.F90:258-302: accept at ci = n+2, or after the trailing NUL at n+3
      fin[static_cast<size_t>(s)] = (s != 0 && (A.out[static_cast<size_t>(s)] || A.out[static_cast<size_t>(TA(s, static_cast<int>(h.cls_nul)))])) ? 1 : 0;
   h.off_finalM = bl.put(fin.data(), fin.size());
   h.off_prefix = bl.put(lit.prefix.data(), lit.prefix.size());
   h.off_suffix = bl.put(lit.suffix.data(), lit.suffix.size());
   h.off_all = bl.put(lit.all.data(), lit.all.size());

   // ---- 7. fast path: <= 8 states per automaton, fused ASCII byte tables -----------------------------------------
   // Candidate-list search == brute-force search on pure-ASCII rows iff the prefix is a NECESSARY, non-self-overlapping
   // beginning of every non-empty match; the suffix is only consulted by the candidate-list driver.
   bool brute_equiv = op == OP_SEARCH && R.ok;   // (a suffix literal without a prefix literal is never consulted: api_internal_m.F90:76-82)
   bool suffix_unproven = false;
   if (brute_equiv && prefilter && has_suffix) {
      // With a suffix literal the driver also (a) gives up when the suffix does not occur, (b) stops at candidates behind its
      // last occurrence (api_internal_m.F90:99-116; one index is a text index, the other a wrapped one).  Neither changes a
      // result when the suffix is a NECESSARY ending of every match and every match is LONGER than the suffix (then the
      // suffix of the match found starts at least one byte behind the match start, which is all the off-by-one needs).
      // Necessity is checked on the NFA walked BACKWARDS from the exit: at each of the last ls positions only the suffix's own
      // symbol -- a singleton class -- leads anywhere, and the entry state (a complete, shorter match) is not met on the way.
      // (round 6: where that proof fails, the kernel checks per row that the match it found ends with the suffix, behind its start: `suffix_check` below)
      suffix_unproven = !suffix_is_necessary_ending();
   }
   // the prefix literal is a NECESSARY beginning of every non-empty match (walking A along it, each state is non-accepting and only
   // the next prefix symbol -- a singleton class -- is live): with it the general engine may skip the reference's brute-force
   // fallback on pure-ASCII rows that do not contain the prefix at all; with border-freeness on top, the tile kernels apply
   bool prefix_necessary = false;
   if (op == OP_SEARCH && prefilter) {
      bool ok = lit.prefix.find('\0') == std::string::npos;
      int q = A.init;
      std::vector<int32_t> codes = decode_chars(lit.prefix);
      for (size_t i = 0; ok && i < codes.size(); ++i) {
         if (i > 0 && A.out[static_cast<size_t>(q)]) ok = false;
         int iv = interval_of(codes[i]);
         int c = cls_of[static_cast<size_t>(iv)];
         if (!(bounds[static_cast<size_t>(iv)] == codes[i] && bounds[static_cast<size_t>(iv) + 1] == codes[i] + 1)) ok = false;
         for (int k = 0; ok && k < nI; ++k)
            if (k != iv && cls_of[static_cast<size_t>(k)] == c) ok = false;
         for (int k = 0; ok && k < ncls; ++k)
            if (k != c && TA(q, k) != 0) ok = false;
         if (ok) {
            q = TA(q, c);
            if (q == 0) ok = false;
         }
      }
      prefix_necessary = ok;
   }
   if (prefix_necessary) h.flags |= FXP_F_PREFIX_NECESSARY;
   // Round 6 (0.6 % of generated patterns on the general kernel, a 20-30 x cliff): where that proof fails -- the prefix is not a necessary
   // beginning (`(}[abc]){2}\d*c{2,}`), or it has a border and no overlap state (`(\t{3}[a-z]){2}`) -- the tile tables still run, and the KERNEL decides per row
   // whether brute force and the candidate list agree.  The reference tries the candidates -- the non-overlapping occurrences of the prefix
   // (utility_m.f90:94-116) -- in order and takes the first with a non-empty match (api_internal_m.F90:108-164); brute force takes the leftmost start s of
   // ANY match.  No candidate before s matches (s is the leftmost start of all), so when s itself is a candidate the two agree: same start, same longest end.
   // s is a candidate when the prefix stands at s and no earlier occurrence overlaps it (a selected one would have swallowed it; an unselected one only
   // makes the test conservative).  Every other row with a hit -- s at the leading NUL, no prefix at s, an overlapping occurrence before s -- goes to the
   // general row procedure inside the same launch, which follows the driver to the letter (it also knows the fallback to brute force when the prefix occurs
   // nowhere).  A row without a hit has no match for the reference either: whatever it finds is a match brute force would have found.  (Suffix literal: only
   // with the suffix proof above -- `brute_equiv` is already false otherwise.)
   // With a suffix literal the driver also gives up when the suffix does not occur in the text and stops at candidates behind its last occurrence
   // (api_internal_m.F90:99-116; the list builder stops collecting behind it as well, utility_m.f90:85-114).  None of that changes the result of a row on which
   // the match found at the candidate s ENDS with the suffix literal at least one byte behind s: the suffix then occurs behind s, so the text index of its last
   // occurrence is >= the wrapped index of s, no candidate up to s is cut off, and every listed occurrence before s ends before the suffix does.  Where the
   // compile-time suffix proof fails the kernel checks exactly that (FXP_F_SUFFIX_CHECK), after the forward pass; rows that fail go to the general procedure.
   bool prefix_check = false, suffix_check = false;
   if (brute_equiv && prefilter && (suffix_unproven || !(prefix_necessary && (border_free(lit.prefix) || overlap_sink)))) {
      suffix_check = suffix_unproven;
      prefix_check = lit.prefix.find('\0') == std::string::npos && lit.prefix.size() <= 32 &&
                     (!suffix_check || (lit.suffix.find('\0') == std::string::npos && lit.suffix.size() <= 32));
      brute_equiv = prefix_check;
   }
   if (prefix_check) h.flags |= FXP_F_PREFIX_CHECK | (suffix_check ? FXP_F_SUFFIX_CHECK : 0u);
   if (overlap_sink) {
      h.flags |= FXP_F_OVERLAP_SINK;
      h.R_inv = static_cast<uint32_t>(R_inv_state);
   }
   // ---- 7. fast path: <= 8 states per automaton, fused byte tables (one v_perm_b32 per input byte) ---------------------
   // `.match.` runs one forward pass of A over the whole row (api_internal_m.F90:258-302): no R, no candidate list, so the
   // tile kernel applies whenever the tables fit; its prefix/suffix gate is evaluated on the row bytes by the kernel.
   const bool is_match = op == OP_MATCH;
   const bool fast = is_match ? A.n <= 8 : (brute_equiv && A.n <= 8 && R.n <= 8);
   // Symbol ids of the fast tables: 0..127 = the ASCII byte itself; 128+c = a multi-byte (or invalid) character of class c
   // (fx_translate rewrites such bytes); 254 = KILL (all-dead row, feeds the end of a row); 255 = SKIP (continuation byte
   // inside a valid character).
   std::vector<uint8_t> fa(256 * 8, 0), fr(256 * 8, 0);
   if (fast) {
      for (int b = 0; b < 128; ++b)
         for (int s = 0; s < 8; ++s) {
            fa[static_cast<size_t>(b) * 8 + s] = s < A.n ? static_cast<uint8_t>(TA(s, ac[static_cast<size_t>(b)])) : 0;
            fr[static_cast<size_t>(b) * 8 + s] = (!is_match && s < R.n) ? static_cast<uint8_t>(TR(s, ac[static_cast<size_t>(b)])) : static_cast<uint8_t>(0);
         }
      h.flags |= FXP_F_FAST_OK;
      const bool utf8 = (is_match || (r_has_skip && !prefilter)) && ncls <= 126;   // id 254 stays an all-dead row (end-of-row kill symbol)
      if (utf8) {
         h.flags |= FXP_F_FAST_UTF8;
         for (int c = 0; c < ncls; ++c)
            for (int s = 0; s < 8; ++s) {
               fa[static_cast<size_t>(128 + c) * 8 + s] = s < A.n ? static_cast<uint8_t>(TA(s, c)) : 0;
               fr[static_cast<size_t>(128 + c) * 8 + s] = (!is_match && s < R.n) ? static_cast<uint8_t>(TR(s, c)) : static_cast<uint8_t>(0);
            }
      }
      // symbol 255 (SKIP: continuation byte inside a character, also the pad behind rows shorter than their 16-byte chunks):
      // identity for A; for R "same W, hit cleared" (a plain identity when R was built SKIP-blind).  254 (KILL) stays all-dead.
      for (int s = 0; s < 8; ++s) {
         fa[255u * 8 + s] = static_cast<uint8_t>(s < A.n ? s : 0);
         const int sk = (is_match || s >= R.n) ? 0 : (r_has_skip ? R.T[static_cast<size_t>(s) * R.ncol + nI] : s);
         fr[255u * 8 + s] = static_cast<uint8_t>(sk);
      }
      h.flags |= FXP_F_RAGGED_OK;
      int accmin = A.n, hitmin = is_match ? 0 : R.n;
      for (int s = A.n - 1; s >= 0 && A.out[static_cast<size_t>(s)]; --s) accmin = s;
      if (!is_match)
         for (int s = R.n - 1; s >= 0 && R.out[static_cast<size_t>(s)]; --s) hitmin = s;
      h.fast_accA_min = static_cast<uint32_t>(accmin);
      h.fast_hitR_min = static_cast<uint32_t>(hitmin);
      h.fast_R_start = h.R_start;
      h.fast_A_init = is_match ? h.M_start : h.A_init;
      uint64_t fm = 0;
      for (int s = 0; s < A.n && s < 8; ++s)
         if (fin[static_cast<size_t>(s)]) fm |= uint64_t(1) << (8 * s);
      h.fast_finalM[0] = static_cast<uint32_t>(fm);
      h.fast_finalM[1] = static_cast<uint32_t>(fm >> 32);
   }
   h.off_fastA = bl.put(fa.data(), fa.size());
   h.off_fastR = bl.put(fr.data(), fr.size());
   // FXP_F_R_LATCH (round 6): R with <= 4 states, doubled -- state s + 4 = "state s, and a hit state has been entered since the latch was last cleared".  From
   // an unlatched state the step goes to fr's destination, latched when that destination is a hit state; from a latched state to fr's destination, latched.
   // The kernels that use it (half-row first pass, span kernel) clear the latch (state & 3) at every 8-byte group's start: the group's last state then says
   // whether any of its eight states was a hit -- the running maximum over them (v_max3_u32, one per two bytes) is gone.
   h.off_fastRL = 0;   // (no latched format: the flag says so)
   if (fast && !is_match && R.n <= 4 && !overlap_sink) {
      std::vector<uint8_t> frl(256 * 8, 0);
      const int hitmin = static_cast<int>(h.fast_hitR_min);
      for (int sym = 0; sym < 256; ++sym)
         for (int s = 0; s < 4; ++s) {
            const int d = fr[static_cast<size_t>(sym) * 8 + s];   // (0 for s >= R.n and for symbols without a row: the dead / start state as in fr)
            frl[static_cast<size_t>(sym) * 8 + s] = static_cast<uint8_t>(d >= hitmin && d < R.n ? d + 4 : d);
            frl[static_cast<size_t>(sym) * 8 + 4 + s] = static_cast<uint8_t>(d + 4);
         }
      h.off_fastRL = bl.put(frl.data(), frl.size());
      h.flags |= FXP_F_R_LATCH;
   }

   // ---- 8. chain tables: any automaton whose class-indexed tables fit 16-bit row offsets (LDS chain kernel) ---------------
   {
      const uint32_t ncols = static_cast<uint32_t>(ncls) + 3, row_bytes = ncols * 2;
      const uint32_t col_skip = static_cast<uint32_t>(ncls), col_kill = static_cast<uint32_t>(ncls) + 1, col_final = static_cast<uint32_t>(ncls) + 2;
      const bool chain = (is_match || brute_equiv) && !fast && ncls <= 126 && static_cast<uint64_t>(A.n) * row_bytes < 65536u &&
                         (is_match || static_cast<uint64_t>(R.n) * row_bytes < 65536u);
      std::vector<uint16_t> cm(256), ctr, cta;
      if (chain) {
         for (uint32_t sym = 0; sym < 256; ++sym) {
            uint32_t col = col_kill;
            if (sym < 128) col = ac[sym];
            else if (sym == 255) col = col_skip;
            else if (sym - 128 < static_cast<uint32_t>(ncls)) col = sym - 128;
            cm[sym] = static_cast<uint16_t>(2 * col);
         }
         ctr.assign(is_match ? 0 : static_cast<size_t>(R.n) * ncols, 0);
         cta.assign(static_cast<size_t>(A.n) * ncols, 0);
         for (int st = 0; !is_match && st < R.n; ++st) {
            for (int c = 0; c < ncls; ++c) ctr[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(TR(st, c) * row_bytes);
            const int sk = r_has_skip ? R.T[static_cast<size_t>(st) * R.ncol + nI] : st;
            ctr[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(sk * row_bytes);
            ctr[static_cast<size_t>(st) * ncols + col_kill] = static_cast<uint16_t>(st * row_bytes);
         }
         for (int st = 0; st < A.n; ++st) {
            for (int c = 0; c < ncls; ++c) cta[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(TA(st, c) * row_bytes);
            cta[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(st * row_bytes);
            cta[static_cast<size_t>(st) * ncols + col_kill] = 0;
            cta[static_cast<size_t>(st) * ncols + col_final] = fin[static_cast<size_t>(st)];
         }
         int accmin = A.n, hitmin = is_match ? 0 : R.n;
         for (int st = A.n - 1; st >= 0 && A.out[static_cast<size_t>(st)]; --st) accmin = st;
         if (!is_match)
            for (int st = R.n - 1; st >= 0 && R.out[static_cast<size_t>(st)]; --st) hitmin = st;
         h.flags |= FXP_F_CHAIN_OK | FXP_F_RAGGED_OK;
         if (is_match || (r_has_skip && !prefilter)) h.flags |= FXP_F_CHAIN_UTF8;
         h.chain_row_bytes = row_bytes;
         h.chain_R_start = h.R_start * row_bytes;
         h.chain_A_init = (is_match ? h.M_start : h.A_init) * row_bytes;
         h.chain_hit_min = static_cast<uint32_t>(hitmin) * row_bytes;
         h.chain_acc_min = static_cast<uint32_t>(accmin) * row_bytes;
         h.chain_TR_bytes = static_cast<uint32_t>(ctr.size() * 2);
         h.chain_TA_bytes = static_cast<uint32_t>(cta.size() * 2);
      }
      h.off_chain_cls = bl.put(cm.data(), cm.size() * 2);
      h.off_chain_TR = bl.put(ctr.data(), ctr.size() * 2);
      h.off_chain_TA = bl.put(cta.data(), cta.size() * 2);
   }

   // ---- 8b. 16-state nibble tables for automata with 9..16 states (same conditions as the 8-state fast tables) ----------------------
   // NIBBLE format: per symbol 8 bytes = 16 nibbles, nibble j = next state of state j (states are plain ids 0..15)
   auto enc16 = [](int i) -> uint8_t { return static_cast<uint8_t>(i); };
   auto min16 = [&](int first_out, int n) -> uint32_t { return first_out >= n ? 0xFFu : enc16(first_out); };   // "no such state": never reached
   auto put_nib = [](std::vector<uint8_t>& t, uint32_t sym, int st, int next) {
      uint8_t& b = t[sym * 8u + static_cast<uint32_t>(st >> 1)];
      b = static_cast<uint8_t>((st & 1) ? ((b & 0x0F) | (next << 4)) : ((b & 0xF0) | (next & 15)));
   };
   {
      const bool w16 = !fast && (is_match ? A.n <= 16 : (brute_equiv && A.n <= 16 && R.n <= 16));
      std::vector<uint8_t> wa(256 * 8, 0), wr(256 * 8, 0);
      if (w16) {
         auto put = [&](uint32_t sym, int c_or_neg, bool skip) {   // one symbol row of both tables; c < 0: all-dead (KILL)
            for (int st = 0; st < 16; ++st) {
               int ta = 0, tr = 0;
               if (skip) {
                  ta = st < A.n ? st : 0;
                  tr = (is_match || st >= R.n) ? 0 : (r_has_skip ? R.T[static_cast<size_t>(st) * R.ncol + nI] : st);
               } else if (c_or_neg >= 0) {
                  ta = st < A.n ? TA(st, c_or_neg) : 0;
                  tr = (!is_match && st < R.n) ? TR(st, c_or_neg) : 0;
               }
               put_nib(wa, sym, st, ta);
               put_nib(wr, sym, st, tr);
            }
         };
         for (uint32_t b = 0; b < 128; ++b) put(b, ac[b], false);
         h.flags |= FXP_F_W16_OK | FXP_F_RAGGED_OK;
         if ((is_match || (r_has_skip && !prefilter)) && ncls <= 126) {
            h.flags |= FXP_F_W16_UTF8;
            for (int c = 0; c < ncls; ++c) put(128u + static_cast<uint32_t>(c), c, false);
         }
         put(255u, 0, true);
         int accmin = A.n, hitmin = is_match ? 0 : R.n;
         for (int st = A.n - 1; st >= 0 && A.out[static_cast<size_t>(st)]; --st) accmin = st;
         if (!is_match)
            for (int st = R.n - 1; st >= 0 && R.out[static_cast<size_t>(st)]; --st) hitmin = st;
         h.w16_acc_min = min16(accmin, A.n);
         h.w16_hit_min = is_match ? 0xFFu : min16(hitmin, R.n);
         h.w16_R_start = enc16(static_cast<int>(h.R_start));
         h.w16_A_init = enc16(static_cast<int>(is_match ? h.M_start : h.A_init));
         uint8_t fm[16] = {0};
         for (int st = 0; st < A.n && st < 16; ++st) fm[st] = fin[static_cast<size_t>(st)];
         std::memcpy(h.w16_finalM, fm, 16);
      }
      h.off_w16A = bl.put(wa.data(), w16 ? wa.size() : 0);
      h.off_w16R = bl.put(wr.data(), w16 ? wr.size() : 0);
   }

   // ---- 9. byte-level chain tables: A and R composed with the UTF-8 decoder (no decode pass on the device) ---------------------
   // Only where the tile kernels' brute-force semantics hold: `.match.`, or a search proven equal to brute force.  With a
   // prefilter literal that proof carries over from pure-ASCII rows to rows of valid canonical UTF-8 -- UTF-8 is
   // self-synchronising, so the byte-level INDEX of the candidate list (api_internal_m.F90:76-104) finds exactly the
   // character-aligned occurrences -- and every other row is an exception row anyway (for these programs they go to the general
   // engine, which follows the candidate-list driver to the letter).
   {
      std::vector<uint16_t> bcm(256, 0), btr, bta;
      // (FXP_F_PREFIX_CHECK programs: no byte-level tables -- the per-row check compares raw bytes of pure-ASCII rows; rows with a byte >= 0x80 go to the general procedure)
      const bool want = (h.flags & (FXP_F_FAST_OK | FXP_F_CHAIN_OK | FXP_F_W16_OK)) != 0 && (is_match || (brute_equiv && !overlap_sink && !prefix_check));
      std::vector<uint8_t> bwa, bwr, b8a;
      if (want) {
         Sig full;
         for (int k = 0; k < nI; ++k) full.emplace_back(static_cast<uint32_t>(bounds[static_cast<size_t>(k)]), cls_of[static_cast<size_t>(k)]);
         sig_normalise(full);
         ClassDfaView va{A.n, [&](int st, int c) { return TA(st, c); }, &A.out, static_cast<int>(is_match ? h.M_start : h.A_init)};
         std::vector<int> fin_i(fin.begin(), fin.end());
         ByteDfa Ab = build_forward_bytes(va, full, fin_i, is_match);
         ByteDfa Rb;
         if (!is_match) {
            ClassDfaView vr{R.n, [&](int st, int c) { return TR(st, c); }, &R.out, R.init};
            Rb = build_reverse_bytes(vr, full);
         }
         if (Ab.ok && (is_match || Rb.ok)) {
            // byte classes: byte values with identical columns in both automata
            std::map<std::vector<int>, int> m;
            std::vector<int> bcls(256), rep;
            for (int b = 0; b < 256; ++b) {
               std::vector<int> col;
               for (int st = 0; st < Ab.d.n; ++st) col.push_back(Ab.d.T[static_cast<size_t>(st) * 256 + b]);
               if (!is_match)
                  for (int st = 0; st < Rb.d.n; ++st) col.push_back(Rb.d.T[static_cast<size_t>(st) * 256 + b]);
               auto it = m.find(col);
               if (it == m.end()) {
                  it = m.emplace(col, static_cast<int>(m.size())).first;
                  rep.push_back(b);
               }
               bcls[static_cast<size_t>(b)] = it->second;
            }
            const uint32_t nbc = static_cast<uint32_t>(m.size()), ncols = nbc + 3, row_bytes = ncols * 2;
            const uint32_t col_skip = nbc, col_kill = nbc + 1, col_final = nbc + 2;
            const uint64_t ta_bytes = static_cast<uint64_t>(Ab.d.n) * row_bytes, tr_bytes = is_match ? 0 : static_cast<uint64_t>(Rb.d.n) * row_bytes;
            if (ta_bytes < 65536u && tr_bytes < 65536u && ta_bytes + tr_bytes <= 48u * 1024u) {
               for (int b = 0; b < 256; ++b) bcm[static_cast<size_t>(b)] = static_cast<uint16_t>(2 * bcls[static_cast<size_t>(b)]);
               bta.assign(static_cast<size_t>(Ab.d.n) * ncols, 0);
               for (int st = 0; st < Ab.d.n; ++st) {
                  for (uint32_t c = 0; c < nbc; ++c)
                     bta[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(Ab.d.T[static_cast<size_t>(st) * 256 + rep[c]] * row_bytes);
                  bta[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(st * row_bytes);
                  bta[static_cast<size_t>(st) * ncols + col_kill] = 0;
                  bta[static_cast<size_t>(st) * ncols + col_final] = static_cast<uint16_t>(Ab.fin[static_cast<size_t>(st)]);
               }
               if (!is_match) {
                  btr.assign(static_cast<size_t>(Rb.d.n) * ncols, 0);
                  for (int st = 0; st < Rb.d.n; ++st) {
                     for (uint32_t c = 0; c < nbc; ++c)
                        btr[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(Rb.d.T[static_cast<size_t>(st) * 256 + rep[c]] * row_bytes);
                     btr[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(st * row_bytes);
                     btr[static_cast<size_t>(st) * ncols + col_kill] = static_cast<uint16_t>(st * row_bytes);
                  }
               }
               int accmin = Ab.d.n, hitmin = is_match ? 0 : Rb.d.n;
               for (int st = Ab.d.n - 1; st >= 0 && Ab.d.out[static_cast<size_t>(st)]; --st) accmin = st;
               if (!is_match)
                  for (int st = Rb.d.n - 1; st >= 0 && Rb.d.out[static_cast<size_t>(st)]; --st) hitmin = st;
               h.flags |= FXP_F_BYTE_DFA;
               // FXP_F_NEEDS_NONASCII: no non-empty match consists of ASCII symbols only (walking A from its initial state over the classes
               // of code points 0..127 -- the NUL sentinels included -- never reaches an accepting state): a row without a byte >= 0x80
               // holds no match, whatever the driver (every reported match is an accepting walk of A over a piece of NUL // text // NUL).
               if (!is_match && !(h.flags & FXP_F_INIT_ACCEPTING)) {
                  std::vector<char> seen(static_cast<size_t>(A.n), 0);
                  std::vector<int> todo{A.init};
                  seen[static_cast<size_t>(A.init)] = 1;
                  bool reach_acc = false;
                  while (!todo.empty() && !reach_acc) {
                     const int st = todo.back();
                     todo.pop_back();
                     for (int b = 0; b < 128 && !reach_acc; ++b) {
                        const int d = TA(st, ac[static_cast<size_t>(b)]);
                        if (d == 0) continue;
                        if (A.out[static_cast<size_t>(d)]) reach_acc = true;
                        if (!seen[static_cast<size_t>(d)]) {
                           seen[static_cast<size_t>(d)] = 1;
                           todo.push_back(d);
                        }
                     }
                  }
                  if (!reach_acc) h.flags |= FXP_F_NEEDS_NONASCII;
               }
               h.byte_n_classes = nbc;
               h.byte_row_bytes = row_bytes;
               h.byte_A_init = static_cast<uint32_t>(Ab.d.init) * row_bytes;
               h.byte_R_start = is_match ? 0u : static_cast<uint32_t>(Rb.d.init) * row_bytes;
               h.byte_acc_min = static_cast<uint32_t>(accmin) * row_bytes;
               h.byte_hit_min = static_cast<uint32_t>(hitmin) * row_bytes;
               h.byte_TA_bytes = static_cast<uint32_t>(bta.size() * 2);
               h.byte_TR_bytes = static_cast<uint32_t>(btr.size() * 2);
               h.byte_inv_A = Ab.inv >= 0 ? static_cast<uint32_t>(Ab.inv) * row_bytes : 0u;
               h.byte_inv_R = is_match ? 0u : static_cast<uint32_t>(Rb.inv) * row_bytes;
               // the same automata in the 16-state v_perm format, indexed by the raw byte
               if (Ab.d.n <= 16 && (is_match || Rb.d.n <= 16)) {
                  bwa.assign(256 * 8, 0);
                  bwr.assign(256 * 8, 0);
                  for (int b = 0; b < 256; ++b)
                     for (int st2 = 0; st2 < 16; ++st2) {
                        put_nib(bwa, static_cast<uint32_t>(b), st2, st2 < Ab.d.n ? Ab.d.T[static_cast<size_t>(st2) * 256 + b] : 0);
                        put_nib(bwr, static_cast<uint32_t>(b), st2, (!is_match && st2 < Rb.d.n) ? Rb.d.T[static_cast<size_t>(st2) * 256 + b] : 0);
                     }
                  h.flags |= FXP_F_BYTE_W16;
                  h.bw16_acc_min = min16(accmin, Ab.d.n);
                  h.bw16_hit_min = is_match ? 0xFFu : min16(hitmin, Rb.d.n);
                  h.bw16_A_init = enc16(Ab.d.init);
                  h.bw16_R_start = is_match ? 0u : enc16(Rb.d.init);
                  h.bw16_inv_A = Ab.inv >= 0 ? enc16(Ab.inv) : 0u;
                  h.bw16_inv_R = is_match ? 0u : enc16(Rb.inv);
                  uint8_t fm[16] = {0};
                  for (int st2 = 0; st2 < Ab.d.n; ++st2) fm[st2] = static_cast<uint8_t>(Ab.fin[static_cast<size_t>(st2)]);
                  std::memcpy(h.bw16_finalM, fm, 16);
                  // searches whose byte-level FORWARD automaton has <= 8 states: the same in the v_perm format (one v_perm_b32 per
                  // byte on the forward pass of a UTF-8 tile; the backward pass keeps the nibble tables)
                  if (!is_match && Ab.d.n <= 8) {
                     b8a.assign(256 * 8, 0);
                     for (int b = 0; b < 256; ++b)
                        for (int st2 = 0; st2 < Ab.d.n; ++st2)
                           b8a[static_cast<size_t>(b) * 8 + static_cast<size_t>(st2)] = static_cast<uint8_t>(Ab.d.T[static_cast<size_t>(st2) * 256 + b]);
                     h.flags |= FXP_F_BYTE_A8;
                     h.b8_A_init = static_cast<uint32_t>(Ab.d.init);
                     h.b8_acc_min = static_cast<uint32_t>(accmin);   // (== Ab.d.n when no state accepts: never reached)
                     // FXP_F_SPEC_FWD: the tile kernels may try the row's FIRST character as the leftmost start with this automaton alone
                     // (api_internal_m.F90:84-88,108-155: the candidates are the leading NUL, then the first character, ...).  Sound when
                     //  * the leading NUL is no start: A dies on it from its initial state;
                     //  * no state of A survives U+FFFF: at a structure error the reference feeds U+FFFF for the bytes of the broken
                     //    sequence (:129-133) and the walk ends there -- as this automaton's does, whose structure errors are dead ends
                     //    (no accept lies between the last character boundary and the error in either);
                     //  * there is no candidate-list driver (its equivalence proof covers canonical UTF-8 only).
                     bool ffff_dead = !prefilter && TA(A.init, static_cast<int>(h.cls_nul)) == 0;
                     for (int st2 = 1; st2 < A.n && ffff_dead; ++st2) ffff_dead = TA(st2, static_cast<int>(h.cls_ffff)) == 0;
                     if (ffff_dead) h.flags |= FXP_F_SPEC_FWD;
                  }
               }
            }
         }
      }
      h.off_b8A = bl.put(b8a.data(), b8a.size());
      h.off_byte_cls = bl.put(bcm.data(), bcm.size() * 2);
      h.off_byte_TR = bl.put(btr.data(), btr.size() * 2);
      h.off_byte_TA = bl.put(bta.data(), bta.size() * 2);
      h.off_bw16A = bl.put(bwa.data(), bwa.size());
      h.off_bw16R = bl.put(bwr.data(), bwr.size());
   }
   return finish(h, bl);
}

Program make_search_literal(const std::string& all) {
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = FXP_MODE_SEARCH_LITERAL;
   h.len_all = static_cast<uint32_t>(all.size());
   h.off_all = bl.put(all.data(), all.size());
   // Tile-kernel tables for INDEX(str, all): the right-to-left pass runs the KMP automaton of the REVERSED literal over raw
   // bytes; it is in its last state exactly at the first byte of an occurrence, and the last such hit seen is the leftmost
   // occurrence.  No forward pass: to = from + len - 1.  (A literal holding a NUL byte would also match the kernel's NUL
   // sentinels: those stay on the general kernel.)
   const int m = static_cast<int>(all.size());
   std::vector<uint8_t> fa(256 * 8, 0), fr(256 * 8, 0);
   std::vector<uint16_t> cm(256, 0), ctr, cta;
   if (m >= 1 && all.find('\0') == std::string::npos) {
      const std::string rev(all.rbegin(), all.rend());
      std::vector<int> delta(static_cast<size_t>(m + 1) * 256, 0);   // KMP automaton of rev: delta[q][b], state m = full match
      {
         auto P = [&](int j) { return static_cast<int>(static_cast<unsigned char>(rev[static_cast<size_t>(j)])); };
         delta[static_cast<size_t>(P(0))] = 1;
         int x = 0;   // state reached on the longest proper border of rev[0..j)
         for (int j = 1; j < m; ++j) {
            for (int b2 = 0; b2 < 256; ++b2) delta[static_cast<size_t>(j) * 256 + b2] = delta[static_cast<size_t>(x) * 256 + b2];
            delta[static_cast<size_t>(j) * 256 + P(j)] = j + 1;
            x = delta[static_cast<size_t>(x) * 256 + P(j)];
         }
         for (int b2 = 0; b2 < 256; ++b2) delta[static_cast<size_t>(m) * 256 + b2] = delta[static_cast<size_t>(x) * 256 + b2];   // overlapping occurrences
      }
      // byte classes: each distinct literal byte, plus "any other byte"
      std::vector<int> cls(256, -1);
      int ncls = 0;
      for (unsigned char ch : all)
         if (cls[ch] < 0) cls[ch] = ncls++;
      const int other = ncls++;
      int other_byte = -1;
      for (int b = 0; b < 256; ++b)
         if (cls[static_cast<size_t>(b)] < 0) {
            cls[static_cast<size_t>(b)] = other;
            if (other_byte < 0) other_byte = b;
         }
      h.flags |= FXP_F_RAW_BYTES;
      if (all.find('\xFF') == std::string::npos) h.flags |= FXP_F_RAGGED_OK;   // pad byte 0xFF cannot advance the literal's automaton
      h.nR = static_cast<uint32_t>(m + 1);
      if (m + 1 <= 8) {
         for (int b = 0; b < 256; ++b)
            for (int q = 0; q <= m; ++q) fr[static_cast<size_t>(b) * 8 + q] = static_cast<uint8_t>(delta[static_cast<size_t>(q) * 256 + b]);
         h.flags |= FXP_F_FAST_OK;
         h.fast_R_start = 0;
         h.fast_hitR_min = static_cast<uint32_t>(m);
         h.fast_accA_min = 8;
      } else if (ncls <= 126 && static_cast<uint64_t>(m + 1) * (ncls + 3) * 2 < 65536u && other_byte >= 0) {
         const uint32_t ncols = static_cast<uint32_t>(ncls) + 3, row_bytes = ncols * 2;
         for (int b = 0; b < 256; ++b) cm[static_cast<size_t>(b)] = static_cast<uint16_t>(2 * cls[static_cast<size_t>(b)]);
         ctr.assign(static_cast<size_t>(m + 1) * ncols, 0);
         std::vector<int> rep(static_cast<size_t>(ncls), other_byte);
         for (unsigned char ch : all) rep[static_cast<size_t>(cls[ch])] = ch;
         for (int q = 0; q <= m; ++q)
            for (int c = 0; c < ncls; ++c)
               ctr[static_cast<size_t>(q) * ncols + c] = static_cast<uint16_t>(delta[static_cast<size_t>(q) * 256 + rep[static_cast<size_t>(c)]] * row_bytes);
         cta.assign(ncols, 0);   // a single dead row: the forward pass is not used
         h.flags |= FXP_F_CHAIN_OK;
         h.n_classes = static_cast<uint32_t>(ncls);
         h.chain_row_bytes = row_bytes;
         h.chain_R_start = 0;
         h.chain_A_init = 0;
         h.chain_hit_min = static_cast<uint32_t>(m) * row_bytes;
         h.chain_acc_min = 0xFFFFFFFFu;
         h.chain_TR_bytes = static_cast<uint32_t>(ctr.size() * 2);
         h.chain_TA_bytes = static_cast<uint32_t>(cta.size() * 2);
      }
   }
   h.off_fastA = bl.put(fa.data(), fa.size());
   h.off_fastR = bl.put(fr.data(), fr.size());
   h.off_chain_cls = bl.put(cm.data(), cm.size() * 2);
   h.off_chain_TR = bl.put(ctr.data(), ctr.size() * 2);
   h.off_chain_TA = bl.put(cta.data(), cta.size() * 2);
   const uint32_t none = bl.put(nullptr, 0);
   h.off_bounds = h.off_bound_cls = h.off_ascii_cls = h.off_TA = h.off_TR = h.off_accA = h.off_hitR = h.off_finalM = none;
   h.off_prefix = h.off_suffix = h.off_cls_page = h.off_cls_pages = none;
   return finish(h, bl);
}

// Wire-format check of a program image that did not come from this compiler (fxamd_program_from_blob): every table must lie
// inside the image with the extent its header fields imply, and every entry a kernel uses as an index (state ids, row
// offsets, class ids, page ids) must stay inside its table -- a handle that passes can be run without reading outside the
// uploaded image.  Returns 0 when the image is sound.
int validate_blob(const uint8_t* b, size_t size) {
   if (size < sizeof(FxpHeader) || (size & 15u) != 0) return 1;
   FxpHeader h;
   std::memcpy(&h, b, sizeof(h));
   if (h.magic != FXP_MAGIC || h.version != FXP_VERSION || h.total_bytes != size) return 2;
   if (h.checksum != blob_checksum(b, size)) return 3;
   if (h.mode > FXP_MODE_MATCH_ENGINE) return 4;
   if (h.mode == FXP_MODE_INVALID) return 0;
   auto inside = [&](uint32_t off, uint64_t bytes) { return off >= sizeof(FxpHeader) && (uint64_t)off + bytes <= size; };
   auto u16 = [&](uint32_t off, uint64_t i) { uint16_t v; std::memcpy(&v, b + off + 2 * i, 2); return v; };
   const uint32_t known = FXP_F_INIT_ACCEPTING | FXP_F_PREFILTER | FXP_F_HAS_SUFFIX | FXP_F_FAST_OK | FXP_F_HAS_R | FXP_F_MATCH_LITERAL |
                          FXP_F_FAST_UTF8 | FXP_F_NFA_SIM | FXP_F_CHAIN_OK | FXP_F_CHAIN_UTF8 | FXP_F_RAW_BYTES | FXP_F_RAGGED_OK | FXP_F_BYTE_DFA |
                          FXP_F_W16_OK | FXP_F_W16_UTF8 | FXP_F_BYTE_W16 | FXP_F_PREFIX_NECESSARY | FXP_F_OVERLAP_SINK | FXP_F_BYTE_A8 | FXP_F_SPEC_FWD | FXP_F_NEEDS_NONASCII |
                          FXP_F_PREFIX_CHECK | FXP_F_SUFFIX_CHECK | FXP_F_R_LATCH;
   if (h.flags & ~known) return 5;
   // chain-format table: rows of (ncls + 3) uint16, entries = row offsets of the same table; the 256-entry map holds 2 * column
   auto chain_ok = [&](uint32_t off_cls, uint32_t off_T, uint32_t T_bytes, uint32_t ncls, uint32_t row_bytes, bool final_col) {
      if (row_bytes != (ncls + 3u) * 2u || T_bytes == 0 || T_bytes % row_bytes != 0 || T_bytes > 65534u) return false;
      if (!inside(off_cls, 512) || !inside(off_T, T_bytes)) return false;
      for (uint32_t i = 0; i < 256; ++i)
         if (u16(off_cls, i) > 2u * (ncls + 1u) || (u16(off_cls, i) & 1u)) return false;
      const uint32_t ncols = ncls + 3u;
      for (uint32_t i = 0; i < T_bytes / 2u; ++i) {
         if (final_col && i % ncols == ncols - 1u) continue;   // FINAL column: a verdict, not an offset
         const uint16_t v = u16(off_T, i);
         if (v >= T_bytes || v % row_bytes != 0) return false;
      }
      return true;
   };
   auto state_ok = [&](uint32_t v, uint32_t T_bytes, uint32_t row_bytes) { return v < T_bytes && v % row_bytes == 0; };
   if (h.mode == FXP_MODE_SEARCH_LITERAL) {
      if (!inside(h.off_all, h.len_all)) return 10;
      if ((h.flags & FXP_F_FAST_OK) && (!inside(h.off_fastA, 2048) || !inside(h.off_fastR, 2048))) return 11;
      if (h.flags & FXP_F_CHAIN_OK) {
         if (!chain_ok(h.off_chain_cls, h.off_chain_TR, h.chain_TR_bytes, h.n_classes, h.chain_row_bytes, false)) return 12;
         if (!chain_ok(h.off_chain_cls, h.off_chain_TA, h.chain_TA_bytes, h.n_classes, h.chain_row_bytes, false)) return 13;
         if (!state_ok(h.chain_R_start, h.chain_TR_bytes, h.chain_row_bytes) || !state_ok(h.chain_A_init, h.chain_TA_bytes, h.chain_row_bytes)) return 14;
      }
      if (h.flags & (FXP_F_BYTE_DFA | FXP_F_W16_OK | FXP_F_BYTE_W16 | FXP_F_NFA_SIM | FXP_F_FAST_UTF8 | FXP_F_CHAIN_UTF8 | FXP_F_OVERLAP_SINK)) return 15;
      return 0;
   }
   // ---- engine modes: class map ----
   const uint32_t nc = h.n_classes;
   if (nc == 0 || nc > 0x7FFFu || h.n_bounds == 0 || h.n_bounds > (1u << 22) || h.n_pages == 0 || h.n_pages > 1024u) return 20;
   if (h.cls_nul >= nc || h.cls_ffff >= nc) return 21;
   if (!inside(h.off_bounds, 4ull * h.n_bounds) || !inside(h.off_bound_cls, 2ull * h.n_bounds) || !inside(h.off_ascii_cls, 256) ||
       !inside(h.off_cls_page, 2048) || !inside(h.off_cls_pages, 128ull * h.n_pages))
      return 22;
   {
      int32_t prev = -1;
      for (uint32_t i = 0; i < h.n_bounds; ++i) {
         int32_t v;
         std::memcpy(&v, b + h.off_bounds + 4ull * i, 4);
         if ((i == 0 && v != 0) || v <= prev) return 23;
         prev = v;
         if (u16(h.off_bound_cls, i) >= nc) return 23;
      }
      for (uint32_t i = 0; i < 128; ++i)
         if (u16(h.off_ascii_cls, i) >= nc) return 24;
      for (uint32_t i = 0; i < 1024; ++i)
         if (u16(h.off_cls_page, i) >= h.n_pages) return 25;
      for (uint64_t i = 0; i < 64ull * h.n_pages; ++i)
         if (u16(h.off_cls_pages, i) >= nc) return 26;
   }
   if (!inside(h.off_prefix, h.len_prefix) || !inside(h.off_suffix, h.len_suffix) || !inside(h.off_all, h.len_all)) return 27;
   if (h.flags & FXP_F_NFA_SIM) {
      const uint64_t N = h.nfa_N, w = h.nfa_words;
      if (w == 0 || N == 0 || N + 1 > 32 * w || h.nfa_entry < 1 || h.nfa_entry > N || h.nfa_exit < 1 || h.nfa_exit > N) return 30;
      if (!inside(h.off_nfa_init, 4 * w) || !inside(h.off_nfa_f0, 4 * w) || !inside(h.off_nfa_rstart, 4 * w)) return 31;
      const uint64_t tb = (uint64_t)nc * (N + 1) * w * 4;
      if (!inside(h.off_nfa_fwd, tb) || !inside(h.off_nfa_rev, tb)) return 32;
      if (h.flags & (FXP_F_FAST_OK | FXP_F_CHAIN_OK | FXP_F_W16_OK | FXP_F_BYTE_DFA | FXP_F_BYTE_W16)) return 33;
      return 0;
   }
   // ---- dense tables of the general engine ----
   const uint32_t nA = h.nA, nR = h.nR;
   if (nA < 2 || nA > 0x7FFFu || nR > 0x7FFFu || h.A_init >= nA || h.M_start >= nA) return 40;
   if (!inside(h.off_TA, 2ull * nA * nc) || !inside(h.off_accA, nA) || !inside(h.off_finalM, nA)) return 41;
   for (uint64_t i = 0; i < (uint64_t)nA * nc; ++i)
      if ((u16(h.off_TA, i) & FXP_STATE_MASK) >= nA) return 42;
   if (h.flags & FXP_F_HAS_R) {
      if (nR < 1 || h.R_start >= nR || !inside(h.off_TR, 2ull * nR * nc) || !inside(h.off_hitR, nR)) return 43;
      for (uint64_t i = 0; i < (uint64_t)nR * nc; ++i)
         if ((u16(h.off_TR, i) & FXP_STATE_MASK) >= nR) return 44;
      if ((h.flags & FXP_F_OVERLAP_SINK) && h.R_inv >= nR) return 45;
   } else if (h.flags & (FXP_F_FAST_OK | FXP_F_CHAIN_OK | FXP_F_W16_OK | FXP_F_BYTE_DFA | FXP_F_OVERLAP_SINK)) {
      if (h.mode != FXP_MODE_MATCH_ENGINE) return 46;   // the tile kernels' search needs R
   }
   if ((h.flags & FXP_F_SUFFIX_CHECK) && (!(h.flags & FXP_F_PREFIX_CHECK) || !(h.flags & FXP_F_HAS_SUFFIX) || h.len_suffix < 1u || h.len_suffix > 32u)) return 79;
   if ((h.flags & FXP_F_PREFIX_CHECK) && (!(h.flags & FXP_F_PREFILTER) || h.mode != FXP_MODE_SEARCH_ENGINE || h.len_prefix < 1u || h.len_prefix > 32u ||
                                          (h.flags & (FXP_F_BYTE_DFA | FXP_F_FAST_UTF8 | FXP_F_CHAIN_UTF8 | FXP_F_W16_UTF8)))) return 78;
   if ((h.flags & FXP_F_FAST_OK) && (!inside(h.off_fastA, 2048) || !inside(h.off_fastR, 2048))) return 50;
   if (h.flags & FXP_F_R_LATCH) {   // the latched format of R: <= 4 base states, entries 0..7, a latched state never unlatches
      if (!(h.flags & FXP_F_FAST_OK) || h.mode != FXP_MODE_SEARCH_ENGINE || (h.flags & FXP_F_OVERLAP_SINK) || nR > 4 || !inside(h.off_fastRL, 2048) || h.fast_hitR_min > 4u ||
          h.fast_R_start > 3u)
         return 80;
      for (uint32_t i = 0; i < 2048; ++i) {
         const uint8_t v = b[h.off_fastRL + i];
         if (v > 7u || ((i & 7u) >= 4u && v < 4u)) return 81;
      }
   }
   if ((h.flags & FXP_F_FAST_UTF8) && (!(h.flags & FXP_F_FAST_OK) || nc > 126)) return 51;
   if (h.flags & FXP_F_CHAIN_OK) {
      if (nc > 126) return 52;
      if (!chain_ok(h.off_chain_cls, h.off_chain_TA, h.chain_TA_bytes, nc, h.chain_row_bytes, true)) return 53;
      if (!state_ok(h.chain_A_init, h.chain_TA_bytes, h.chain_row_bytes)) return 54;
      if (h.mode != FXP_MODE_MATCH_ENGINE) {
         if (!chain_ok(h.off_chain_cls, h.off_chain_TR, h.chain_TR_bytes, nc, h.chain_row_bytes, true)) return 55;
         if (!state_ok(h.chain_R_start, h.chain_TR_bytes, h.chain_row_bytes)) return 56;
      } else if (h.chain_TR_bytes != 0 && !inside(h.off_chain_TR, h.chain_TR_bytes)) return 57;
   } else if (h.flags & FXP_F_CHAIN_UTF8) return 58;
   if (h.flags & FXP_F_W16_OK) {
      if (!inside(h.off_w16A, 2048) || !inside(h.off_w16R, 2048)) return 60;
   } else if (h.flags & FXP_F_W16_UTF8) return 61;
   if (h.flags & FXP_F_BYTE_DFA) {
      const uint32_t bc = h.byte_n_classes;
      if (bc == 0 || bc > 256u) return 62;
      if (!chain_ok(h.off_byte_cls, h.off_byte_TA, h.byte_TA_bytes, bc, h.byte_row_bytes, true)) return 63;
      if (!state_ok(h.byte_A_init, h.byte_TA_bytes, h.byte_row_bytes)) return 64;
      if (h.mode != FXP_MODE_MATCH_ENGINE) {
         if (!chain_ok(h.off_byte_cls, h.off_byte_TR, h.byte_TR_bytes, bc, h.byte_row_bytes, true)) return 65;
         if (!state_ok(h.byte_R_start, h.byte_TR_bytes, h.byte_row_bytes)) return 66;
      } else if (h.byte_TR_bytes != 0 && !inside(h.off_byte_TR, h.byte_TR_bytes)) return 67;
      if ((h.flags & FXP_F_BYTE_W16) && (!inside(h.off_bw16A, 2048) || !inside(h.off_bw16R, 2048))) return 68;
      if (h.flags & FXP_F_BYTE_A8) {
         if (!(h.flags & FXP_F_BYTE_W16) || h.mode == FXP_MODE_MATCH_ENGINE || !inside(h.off_b8A, 2048)) return 70;
         if (h.b8_A_init >= 8u || h.b8_acc_min > 8u) return 71;
         const uint8_t* t = b + h.off_b8A;
         for (uint32_t i = 0; i < 2048u; ++i)
            if (t[i] >= 8u) return 72;
         if (h.flags & FXP_F_SPEC_FWD) {   // the claim is checked against the tables: the NUL kills the initial state, U+FFFF kills every state
            if ((h.flags & FXP_F_PREFILTER) || t[h.b8_A_init] != 0u) return 73;
            for (uint32_t st = 1; st < h.nA; ++st)
               if ((u16(h.off_TA, (uint64_t)st * h.n_classes + h.cls_ffff) & FXP_STATE_MASK) != 0u) return 74;
         }
      } else if (h.flags & FXP_F_SPEC_FWD) return 75;
      if (h.flags & FXP_F_NEEDS_NONASCII) {   // the claim is checked against A: no accepting state is reachable over ASCII classes
         if (h.mode != FXP_MODE_SEARCH_ENGINE || (h.flags & FXP_F_INIT_ACCEPTING) || h.A_init >= h.nA) return 76;
         std::vector<char> seen(h.nA, 0);
         std::vector<uint32_t> todo{h.A_init};
         seen[h.A_init] = 1;
         while (!todo.empty()) {
            const uint32_t st = todo.back();
            todo.pop_back();
            for (uint32_t c = 0; c < 128; ++c) {
               const uint16_t e = u16(h.off_TA, (uint64_t)st * h.n_classes + u16(h.off_ascii_cls, c));
               const uint32_t d = e & FXP_STATE_MASK;
               if (d == 0) continue;
               if (e & FXP_FLAG_BIT) return 77;
               if (!seen[d]) {
                  seen[d] = 1;
                  todo.push_back(d);
               }
            }
         }
      }
   } else if (h.flags & (FXP_F_BYTE_W16 | FXP_F_BYTE_A8 | FXP_F_SPEC_FWD | FXP_F_NEEDS_NONASCII)) return 69;
   return 0;
}

std::string pattern_text(const std::string& pattern, int op) {
   std::string buff;
   if (op == OP_SEARCH) {
      buff = f_trim(pattern);   // forgex.F90:95,260
   } else {
      // forgex.F90:182-190 with utility_m.f90:23-53
      std::string adj = f_adjustl(pattern);
      bool caret = !adj.empty() && adj[0] == '^';
      buff = caret ? pattern.substr(1) : pattern;
      std::string tr = f_trim(pattern);
      bool dollar = !tr.empty() && tr[tr.size() - 1] == '$';
      if (dollar) {
         int n = f_len_trim(pattern) - 1;
         if (n < 0) n = 0;
         buff = buff.substr(0, std::min(static_cast<size_t>(n), buff.size()));
      }
   }
   return buff;
}

Program compile(const std::string& pattern, int op, const Limits& lim) {
   const std::string buff = pattern_text(pattern, op);
   Tree tree;
   tree.build(buff);
   if (!tree.is_valid) return invalid_program(tree.code);
   Literals lit = extract_literal(tree);
   if (op == OP_SEARCH && !f_eq(lit.all, "")) return make_search_literal(lit.all);   // forgex.F90:111-130, :281-307
   Nfa nfa = build_nfa(tree, lim.max_nfa_states);
   return compile_from_nfa(nfa, lit, op, lim);
}

}   // namespace fxc
