// range_list_compat.cpp - the reference's RL_Tree API (src/range_list.h:150-162) as exported by libfastq_gpu.so.
//
// A container API like hash.h's: host memory, host callers (fastq_tests.c, bam_umi_count.c).  The bulk GPU path of
// bam_umi_count does not go through it (fqg_umi_count replays the tree's behaviour on the device, fqg_rl_sim.h);
// this file exists so that a program written against range_list.h links against this library unchanged.
//
// Behaviour-exact with the reference's implementation (src/range_list.c), including what makes it differ from a
// set: new_node's shift that moves nothing when exactly one node lies behind the insertion point (:287-301,
// :338-339), rl_all keeping every node but the root's quadrants (:187-198), the node count refreshed with the
// child's interval (:485).  Differential test: tests/test_rl_sim.py::test_compat_range_list_matches_reference_source.
//
// Layout (src/range_list.h:35-41, 96-103): one array of 16-bit nodes in pre-order.  Inner node: quadrant q
// (1..4) in bits 2(q-1)..2(q-1)+1 - 0 out, 1 not part of the range, 2 partially in (a child node follows),
// 3 all in - and the number of nodes of the subtree in bits 8..15, 255 = "more, count them".  Leaf: 16 numbers.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fastq_gpu_compat.h"

namespace {

typedef unsigned long NUM;
enum { kOut = 0, kIgnore = 1, kPart = 2, kAll = 3 };
const NUM kLeaf = 16, kFan = 4;

inline NUM min_num(NUM a, NUM b) { return a < b ? a : b; }
inline NUM narrow(NUM w) { return w <= kLeaf * kFan ? kLeaf : w / kFan + w % kFan; }  // NEXT_INTERVAL
inline bool leaf_width(NUM w) { return w <= kLeaf; }
inline unsigned on_bits(unsigned n) { return n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u; }  // active_bits[n - 1]

inline int quad(const RL_Tree* t, NUM node, int q) {
  if (q < 1 || q > 4) {
    fprintf(stderr, "ERROR: quadrant_status: invalid quadrant(%d)\n", q);
    return 0;
  }
  return (t->root[node].leaf >> (2 * (q - 1))) & 3;
}
inline void set_quad(RL_Tree* t, NUM node, int q, int st) {
  if (q < 1 || q > 4) {
    fprintf(stderr, "ERROR: set_quadrant: invalid quadrant %d(%d)\n", q, st);
    return;
  }
  unsigned short& v = t->root[node].leaf;
  v = (unsigned short)((v & ~(3u << (2 * (q - 1)))) | ((unsigned)st << (2 * (q - 1))));
}
inline unsigned subnodes(const RL_Tree* t, NUM node) { return t->root[node].leaf >> 8; }
inline void set_subnodes(RL_Tree* t, NUM node, unsigned c) {
  unsigned short& v = t->root[node].leaf;
  v = (unsigned short)((v & 0xFF) | (c << 8));
}
inline NUM root_width(const RL_Tree* t) { return t->root_i * kFan; }
inline NUM quad_width(const RL_Tree* t, NUM w) { return t->range_max <= w ? t->root_i : narrow(w); }

unsigned tree_size(const RL_Tree* t, NUM node, NUM w) {  // src/range_list.c:566-593
  if (leaf_width(w)) return 1;
  if (subnodes(t, node) != 255) return subnodes(t, node);
  unsigned c = 1;
  const NUM cw = narrow(w);
  for (int q = 1; q <= 4; ++q)
    if (quad(t, node, q) == kPart) c += tree_size(t, node + c, cw);
  return c;
}

int location(const RL_Tree* t, NUM node, int q, NUM w) {  // get_location, :375-408
  if (q == 1 || leaf_width(w)) return 1;
  int c = 1;
  if (w <= kLeaf * kFan && w > kLeaf) {
    for (int i = 1; i < q; ++i)
      if (quad(t, node, i) == kPart) ++c;
    return c;
  }
  const NUM cw = quad_width(t, w);
  NUM at = node + 1;
  for (int i = 1; i != q && i <= 4; ++i)
    if (quad(t, node, i) == kPart) {
      const int s = (int)tree_size(t, at, cw);
      at += (NUM)s;
      c += s;
    }
  return c;
}

void which_quadrant(const RL_Tree* t, NUM number, NUM w, NUM first, short* q, NUM* lo, NUM* hi) {  // :266-277
  const NUM qw = quad_width(t, w);
  const int i = (int)((number - first) / qw + 1);
  *hi = first - 1 + qw * (NUM)i;
  *q = (short)i;
  *lo = *hi - qw + 1;
}

NUM make_node(RL_Tree* t, NUM father, short q, NUM father_w, NUM lo, NUM hi, STATUS status) {  // new_node, :325-372
  const NUM w = narrow(father_w);
  const NUM at = father + (NUM)location(t, father, q, father_w);
  if (t->mem_alloc != 0) {
    if (t->mem_alloc < (t->size + 1) * sizeof(RL_Node)) {
      RL_Node* p = (RL_Node*)realloc(t->root, (t->size + 2) * sizeof(RL_Node));
      if (!p) {
        fprintf(stderr, "Fatal error:range_list: Unable to allocate memory");
        exit(1);
      }
      t->root = p;
      t->mem_alloc = (t->size + 2) * sizeof(RL_Node);
    }
    // shift_right(tree, at, size - 1 - at): nothing moves unless MORE than one node lies at / behind `at`
    const long behind = (long)(t->size - 1 - at);
    if (behind > 0)
      for (long n = (long)at + behind; n >= (long)at; --n) t->root[n + 1].leaf = t->root[n].leaf;
  }
  set_quad(t, father, q, kPart);
  if (status == IN) {
    t->root[at].leaf = 0;
    if (!leaf_width(w)) {
      set_subnodes(t, at, 1);
      for (short k = 2; k <= 4; ++k)
        if (min_num(hi, t->range_max) < lo + narrow(w) * (NUM)(k - 1)) set_quad(t, at, k, kIgnore);
    }
  } else {
    t->root[at].leaf = (unsigned short)on_bits((unsigned)min_num(16, t->range_max - lo + 1));
    if (!leaf_width(w)) {
      t->root[at].leaf = (unsigned short)((t->root[at].leaf & 0xFF00u) | 0xFFu);  // the four quadrants all in
      set_subnodes(t, at, 1);
      for (short k = 2; k <= 4; ++k)
        if (min_num(hi, t->range_max) < lo + narrow(w) * (NUM)(k - 1)) set_quad(t, at, k, kIgnore);
    }
  }
  t->size++;
  return at;
}

long put(RL_Tree* t, NUM number, NUM node, NUM first, NUM w, STATUS status) {  // set_in, :417-496
  long before = (long)t->size;
  if (leaf_width(w)) {
    unsigned n = (unsigned)(number - first);
    char* bytes = (char*)&t->root[node];  // set_num_bit, :597-607
    if (n >= 8) {
      ++bytes;
      n -= 8;
    }
    if (status == IN) *bytes |= (char)(1 << n);
    else *bytes &= (char)~(1 << n);
    return 0;
  }
  short q;
  NUM lo, hi, next;
  which_quadrant(t, number, w, first, &q, &lo, &hi);
  const int st = quad(t, node, q);
  if (status == IN) {
    if (st == kOut) next = make_node(t, node, q, w, lo, hi, status);
    else if (st == kAll) return 0;
    else next = node + (NUM)location(t, node, q, w);
  } else {
    if (st == kAll) next = make_node(t, node, q, w, lo, hi, status);
    else if (st == kOut) return 0;
    else next = node + (NUM)location(t, node, q, w);
  }
  const NUM cw = hi - lo + 1;
  put(t, number, next, lo, cw, status);
  const long added = (long)t->size - before;
  NUM c;
  if (subnodes(t, node) == 255) c = tree_size(t, node, cw);
  else c = (NUM)(added + (long)subnodes(t, node));
  set_subnodes(t, node, c > 254 ? 255u : (unsigned)c);
  return added;
}

bool has(const RL_Tree* t, NUM number, NUM node, NUM first, NUM w) {  // in_tree, :664-690
  for (;;) {
    if (leaf_width(w)) {
      unsigned n = (unsigned)(number - first);
      const char* bytes = (const char*)&t->root[node];
      if (n >= 8) {
        ++bytes;
        n -= 8;
      }
      return ((*bytes) & (1 << n)) != 0;
    }
    short q;
    NUM lo, hi;
    which_quadrant(t, number, w, first, &q, &lo, &hi);
    if (quad(t, node, q) != kPart) return quad(t, node, q) == kAll;  // (asked twice, like the reference: :682-688)
    node += (NUM)location(t, node, q, w);
    first = lo;
    w = hi - lo + 1;
  }
}

bool leaf_bit(const RL_Tree* t, NUM node, unsigned n) {
  const char* bytes = (const char*)&t->root[node];
  if (n >= 8) {
    ++bytes;
    n -= 8;
  }
  return ((*bytes) & (1 << n)) != 0;
}

void show_leaf(const RL_Tree* t, NUM node, NUM first) {  // display_leaf, :698-708
  printf("|");
  for (unsigned i = 0; i < kLeaf; ++i)
    if (leaf_bit(t, node, i)) printf(",%lu", first + i);
    else printf(",.");
  printf("|");
}

void show(const RL_Tree* t, NUM node, NUM first, NUM w, NUM max) {  // idisplay_tree, :751-785
  if (leaf_width(w)) {
    show_leaf(t, node, first);
    return;
  }
  const NUM cw = narrow(w);
  for (short q = 1; q <= 4; ++q) {
    const NUM f2 = first + (NUM)(q - 1) * cw;
    const NUM qmax = min_num(first + cw * (NUM)q - 1, max);
    switch (quad(t, node, q)) {
      case kPart: {
        const NUM next = node + (NUM)location(t, node, q, w);
        if (leaf_width(cw)) show_leaf(t, next, f2);
        else show(t, next, f2, cw, qmax);
        break;
      }
      case kAll:
        printf(",[%lu-%lu]", f2, min_num(f2 + cw - 1, max));
        break;
      case kIgnore:
        break;
      default:
        printf(",]%lu-%lu[", f2, min_num(t->range_max, f2 + cw - 1));
    }
  }
}

NUM next_at_least(const RL_Tree* t, NUM node, NUM first, NUM w, NUM max, NUM min) {  // next_min, :806-846
  if (min > t->range_max) return 0;
  if (leaf_width(w)) {
    const NUM top = min_num(max, t->range_max);
    for (NUM n = first < min ? min : first; n <= top; ++n)
      if (leaf_bit(t, node, (unsigned)(n - first))) return n;
    return 0;
  }
  const NUM cw = narrow(w);
  for (short q = 1; q <= 4; ++q) {
    const NUM f2 = first + (NUM)(q - 1) * cw;
    const NUM qmax = min_num(first + cw * (NUM)q - 1, max);
    const int st = quad(t, node, q);
    if (st == kPart) {
      const NUM found = next_at_least(t, node + (NUM)location(t, node, q, w), f2, cw, qmax, min);
      if (found > 0) return found;
    } else if (st == kAll) {
      if (min <= qmax && min >= f2) return min;
      if (min < f2) return f2;
    }
  }
  return 0;
}

}  // namespace

extern "C" {

RL_Tree* new_rl(NUM max_size) {  // src/range_list.c:90-127; root_intervals :927-939
  if (max_size < 2) max_size = 2;
  RL_Tree* t = (RL_Tree*)malloc(sizeof(RL_Tree));
  if (!t) return nullptr;
  t->range_max = max_size;
  NUM w = kLeaf;
  if (max_size > kLeaf * kFan) {
    NUM j = kFan;
    for (;;) {
      w = kLeaf * j;
      if (w * kFan >= max_size) break;
      j *= kFan;
    }
  }
  t->root_i = w;
  if (t->root_i * kFan < t->range_max) {
    t->root_i = t->root_i * kFan;
    printf("%lu---->>%lu\n", t->range_max, t->root_i);
  }
  t->root = (RL_Node*)calloc(1, sizeof(RL_Node));
  t->size = 1;
  t->mem_alloc = sizeof(RL_Node);
  t->root[0].leaf = 0;
  set_subnodes(t, 0, 1);
  const NUM qi = quad_width(t, max_size);
  for (short q = 2; q <= 4; ++q)
    if (max_size < qi * (NUM)(q - 1) + 1) set_quad(t, 0, q, kIgnore);
  return t;
}

RL_Tree* copy_rl(RL_Tree* tree) {  // :132-152
  RL_Tree* t = (RL_Tree*)malloc(sizeof(RL_Tree));
  RL_Node* nodes = (RL_Node*)calloc(tree->size, sizeof(RL_Node));
  if (!t) {
    printf("new==NULL");
    return nullptr;
  }
  if (!nodes) {
    printf("buf_ptr==NULL---%lu", tree->size);
    return nullptr;
  }
  memcpy(t, tree, sizeof(RL_Tree));
  memcpy(nodes, tree->root, tree->size * sizeof(RL_Node));
  t->root = nodes;
  t->mem_alloc = tree->size * sizeof(RL_Node);
  return t;
}

void free_rl(RL_Tree* range) {  // :157-164
  if (range->mem_alloc != 0) free(range->root);
  free(range);
}

RL_Tree* set_in_rl(RL_Tree* tree, NUM number, STATUS status) {  // :169-183
  if (status != IN && status != OUT) {
    printf("set_in: invalid number status %d\n", status);
    exit(1);
  }
  if (number > 0 && number <= tree->range_max) put(tree, number, 0, 1, root_width(tree), status);
  return tree;
}

void rl_all(RL_Tree* tree, STATUS status) {  // :187-198
  for (int q = 1; q <= 4; ++q)
    if (quad(tree, 0, q) != kIgnore) set_quad(tree, 0, q, status == IN ? kAll : kOut);
  tree->size = 1;
}

BOOLEAN in_rl(RL_Tree* tree, NUM number) {  // :203-207 (`number < 1 && number > range_max` never holds)
  return has(tree, number, 0, 1, root_width(tree)) ? 1 : 0;
}

BOOLEAN freeze_rl(RL_Tree* range) {  // :212-221
  const NUM s = range->size * sizeof(RL_Node);
  if (s < range->mem_alloc) {
    range->root = (RL_Node*)realloc(range->root, s);
    range->mem_alloc = s;
  }
  return 1;
}

RL_Tree* minus_rl(RL_Tree* range1, RL_Tree* range2) {  // :226-231 (the subtraction itself is commented out there)
  if (range1->range_max != range2->range_max) return nullptr;
  return range1;
}

NUM rl_next_in_bigger(RL_Tree* tree, NUM min) {  // :236-241
  if (tree == nullptr) fprintf(stdout, "!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!%lu\n", min);
  return next_at_least(tree, 0, 1, root_width(tree), tree->range_max, min + 1);
}

void display_tree(RL_Tree* tree) {  // :713-747
  printf("Size:%lu -[1,%lu]\n", tree->size, tree->range_max);
  const NUM qi = root_width(tree) / kFan;
  NUM top = 0;
  for (int q = 1; q <= 4; ++q) {
    top += qi;
    const NUM first = top - qi + 1;
    switch (quad(tree, 0, q)) {
      case kPart:
        show(tree, (NUM)location(tree, 0, q, qi * kFan), first, qi, top);
        break;
      case kAll:
        printf(",[%lu-%lu]", first, min_num(top, tree->range_max));
        break;
      case kIgnore:
        break;
      default:
        printf(",]%lu-%lu[", first, min_num(top, tree->range_max));
    }
  }
  printf("\n");
}

}  // extern "C"
