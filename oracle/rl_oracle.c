/* TEST INFRASTRUCTURE, not product code.
 *
 * Restatement of the integer set the reference's bam_umi_count keeps per (cell, gene): the RL_Tree of
 * reference src/range_list.c (0.25.3) - as it BEHAVES, not as src/range_list.h:150-162 documents it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 *
 * The reference stores a 4-ary tree over [1..range_max] in ONE array of 16-bit nodes in pre-order
 * (src/range_list.h:35-41,96-103).  An inner node = four 2-bit quadrant states (bits 0-7, quadrant 1 in
 * bits 0-1) + an 8-bit count of the nodes of its subtree, saturating at 255 (bits 8-15); a leaf = a
 * 16-number bitmap.  Three things make it differ from a set, all restated here on purpose:
 *   (1) new_node (src/range_list.c:325-349) opens a gap for the new node with shift_right (:287-301), which
 *       returns without moving anything when exactly ONE node lies at/after the insertion point
 *       (nnodes == 0): the new node overwrites the last node of the array, the array still grows by one,
 *       and its new last slot keeps whatever the memory held;
 *   (2) rl_all(OUT) (:187-198) resets the root's quadrants and size = 1 but neither the root's node count
 *       nor the nodes behind it, so (1) can make stale nodes of earlier contents live again;
 *   (3) set_in (:420-496) refreshes a saturated node count with tree_size(node, CHILD interval) (:485), i.e.
 *       one level short.
 * Slots of the array the reference never wrote hold heap bytes; here they read as 0 and every such read is
 * counted in `undefined_reads` (the reference's result is then not a function of its input).  Writes past
 * the end of the array (the reference corrupts its heap) are counted in `wild_writes`.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint16_t *node;   /* the array: node[0] is the root */
    uint8_t *ever;    /* 1 = the reference has written this slot at some time */
    size_t cap;       /* slots we hold (grows as needed; the reference reallocs, src/range_list.c:336-346) */
    uint64_t size;    /* live nodes (RL_Tree.size) */
    uint64_t range_max, root_q; /* root_q = RL_Tree.root_i: width of a root quadrant */
    uint64_t undefined_reads, wild_writes, overwrites;
} orl_tree;

enum { Q_OUT = 0, Q_IGNORE = 1, Q_PART = 2, Q_ALL = 3 };

static uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* width of the quadrants of a node of width w below the root (NEXT_INTERVAL, src/range_list.h:72) */
static uint64_t child_width(uint64_t w) { return w <= 64 ? 16 : w / 4 + w % 4; }
static int is_leaf(uint64_t w) { return w <= 16; }

/* quadrant width of THIS node: the root uses root_q (quadrant_interval, src/range_list.c:256-263) */
static uint64_t quad_width(const orl_tree *t, uint64_t w) { return t->range_max <= w ? t->root_q : child_width(w); }

static void need(orl_tree *t, uint64_t idx) {
    if (idx < t->cap) return;
    size_t nc = t->cap ? t->cap : 16;
    while (nc <= idx) nc *= 2;
    t->node = (uint16_t *)realloc(t->node, nc * sizeof(uint16_t));
    t->ever = (uint8_t *)realloc(t->ever, nc);
    memset(t->node + t->cap, 0, (nc - t->cap) * sizeof(uint16_t));
    memset(t->ever + t->cap, 0, nc - t->cap);
    t->cap = nc;
}
static uint16_t rd(orl_tree *t, uint64_t idx) {
    need(t, idx);
    if (!t->ever[idx]) t->undefined_reads++;
    return t->node[idx];
}
static void wr(orl_tree *t, uint64_t idx, uint16_t v) {
    need(t, idx);
    t->node[idx] = v;
    t->ever[idx] = 1;
}
static int quad_get(orl_tree *t, uint64_t idx, int q) { return (rd(t, idx) >> (2 * (q - 1))) & 3; }
static void quad_set(orl_tree *t, uint64_t idx, int q, int st) {
    uint16_t v = rd(t, idx);
    if (q < 1 || q > 4) return; /* set_quadrant only complains (src/range_list.c:626-628) */
    wr(t, idx, (uint16_t)((v & ~(3u << (2 * (q - 1)))) | ((unsigned)st << (2 * (q - 1)))));
}
static unsigned count_get(orl_tree *t, uint64_t idx) { return rd(t, idx) >> 8; }
static void count_set(orl_tree *t, uint64_t idx, unsigned c) { wr(t, idx, (uint16_t)((rd(t, idx) & 0xFF) | (c << 8))); }

/* nodes of the subtree at idx if it is a node of width w (tree_size, src/range_list.c:566-593) */
static unsigned subtree_nodes(orl_tree *t, uint64_t idx, uint64_t w) {
    if (is_leaf(w)) return 1;
    unsigned c = count_get(t, idx);
    if (c != 255) return c;
    c = 1;
    for (int q = 1; q <= 4; q++)
        if (quad_get(t, idx, q) == Q_PART) c += subtree_nodes(t, idx + c, child_width(w));
    return c;
}

/* distance from a node to the child of its quadrant q (get_location, src/range_list.c:375-408) */
static int child_offset(orl_tree *t, uint64_t idx, int q, uint64_t w) {
    if (q == 1 || is_leaf(w)) return 1;
    int c = 1;
    if (w <= 64) { /* the children are leaves */
        for (int i = 1; i < q; i++)
            if (quad_get(t, idx, i) == Q_PART) c++;
        return c;
    }
    uint64_t cw = quad_width(t, w), at = idx + 1;
    for (int i = 1; i != q && i <= 4; i++)
        if (quad_get(t, idx, i) == Q_PART) {
            int s = (int)subtree_nodes(t, at, cw);
            at += (uint64_t)s;
            c += s;
        }
    return c;
}

/* which quadrant of a node starting at `first` holds `number` (number_quadrant, src/range_list.c:266-277) */
static void locate(const orl_tree *t, uint64_t number, uint64_t w, uint64_t first, int *q, uint64_t *lo, uint64_t *hi) {
    uint64_t qw = quad_width(t, w);
    int i = (int)((number - first) / qw + 1);
    *hi = first - 1 + qw * (uint64_t)i;
    *lo = *hi - qw + 1;
    *q = (short)i;
}

/* new_node with status IN (src/range_list.c:325-372) */
static uint64_t open_node(orl_tree *t, uint64_t father, int q, uint64_t father_w, uint64_t lo, uint64_t hi) {
    uint64_t w = child_width(father_w);
    uint64_t at = father + (uint64_t)child_offset(t, father, q, father_w);
    long behind = (long)(t->size - 1 - at); /* nodes after the insertion point, as the reference counts them */
    if (behind > 0) {
        for (long n = (long)at + behind; n >= (long)at; n--) wr(t, (uint64_t)n + 1, rd(t, (uint64_t)n));
    } else if (behind == 0) {
        t->overwrites++; /* defect (1): node[at], the last node, is about to be lost */
    } else if (at > t->size) {
        t->wild_writes++; /* the reference writes beyond what it allocated for size + 1 nodes */
    }
    quad_set(t, father, q, Q_PART);
    wr(t, at, 0);
    if (!is_leaf(w)) {
        uint16_t v = 1u << 8;
        for (int k = 2; k <= 4; k++)
            if (min_u64(hi, t->range_max) < lo + child_width(w) * (uint64_t)(k - 1)) v |= (uint16_t)(Q_IGNORE << (2 * (k - 1)));
        wr(t, at, v);
    }
    t->size++;
    return at;
}

/* set_in with status IN (src/range_list.c:417-496); returns the nodes it added */
static long insert_below(orl_tree *t, uint64_t number, uint64_t idx, uint64_t first, uint64_t w) {
    long before = (long)t->size;
    if (is_leaf(w)) {
        wr(t, idx, (uint16_t)(rd(t, idx) | (1u << (number - first))));
        return 0;
    }
    int q;
    uint64_t lo, hi, next;
    locate(t, number, w, first, &q, &lo, &hi);
    int st = quad_get(t, idx, q);
    if (st == Q_OUT) next = open_node(t, idx, q, w, lo, hi);
    else if (st == Q_ALL) return 0;
    else next = idx + (uint64_t)child_offset(t, idx, q, w);
    uint64_t cw = hi - lo + 1;
    insert_below(t, number, next, lo, cw);
    long added = (long)t->size - before;
    uint64_t c;
    if (count_get(t, idx) == 255) c = subtree_nodes(t, idx, cw); /* defect (3): the child's width */
    else c = (uint64_t)(added + (long)count_get(t, idx));
    count_set(t, idx, c > 254 ? 255u : (unsigned)c);
    return added;
}

static int member_below(orl_tree *t, uint64_t number, uint64_t idx, uint64_t first, uint64_t w) {
    for (;;) { /* in_tree, src/range_list.c:664-690 */
        if (is_leaf(w)) return (rd(t, idx) >> (number - first)) & 1;
        int q;
        uint64_t lo, hi;
        locate(t, number, w, first, &q, &lo, &hi);
        int st = quad_get(t, idx, q);
        if (st == Q_ALL) return 1;
        if (st != Q_PART) return 0;
        idx += (uint64_t)child_offset(t, idx, q, w);
        first = lo;
        w = hi - lo + 1;
    }
}

/* ---- what the tests call ---- */
orl_tree *orl_new(uint64_t max_size) { /* new_rl, src/range_list.c:90-127; root_intervals :927-939 */
    orl_tree *t = (orl_tree *)calloc(1, sizeof *t);
    if (max_size < 2) max_size = 2;
    t->range_max = max_size;
    uint64_t rq = 16;
    if (max_size > 64) {
        uint64_t j = 4;
        for (;;) {
            rq = 16 * j;
            if (rq * 4 >= max_size) break;
            j *= 4;
        }
    }
    if (rq * 4 < max_size) rq *= 4;
    t->root_q = rq;
    t->size = 1;
    uint16_t v = 1u << 8;
    for (int k = 2; k <= 4; k++)
        if (max_size < rq * (uint64_t)(k - 1) + 1) v |= (uint16_t)(Q_IGNORE << (2 * (k - 1)));
    wr(t, 0, v);
    return t;
}
void orl_free(orl_tree *t) {
    if (!t) return;
    free(t->node);
    free(t->ever);
    free(t);
}
void orl_insert(orl_tree *t, uint64_t number) { /* set_in_rl(.., IN), src/range_list.c:169-183 */
    if (number > 0 && number <= t->range_max) insert_below(t, number, 0, 1, t->root_q * 4);
}
int orl_member(orl_tree *t, uint64_t number) { /* in_rl, :203-207 (its range test can never fail) */
    return member_below(t, number, 0, 1, t->root_q * 4);
}
void orl_all_out(orl_tree *t) { /* rl_all(.., OUT), :187-198 */
    for (int q = 1; q <= 4; q++)
        if (quad_get(t, 0, q) != Q_IGNORE) quad_set(t, 0, q, Q_OUT);
    t->size = 1;
}
uint64_t orl_size(const orl_tree *t) { return t->size; }
uint64_t orl_undefined_reads(const orl_tree *t) { return t->undefined_reads; }
uint64_t orl_wild_writes(const orl_tree *t) { return t->wild_writes; }
uint64_t orl_overwrites(const orl_tree *t) { return t->overwrites; }
uint16_t orl_node(orl_tree *t, uint64_t idx) { return idx < t->cap ? t->node[idx] : 0; }

/* Replay of process_entry's set decisions (src/bam_umi_count.c:478-502) for a whole record stream.
 * tree_of[i] = which tree record i touches (sorted mode: the feature id - one tree per feature for the
 * whole file; unsorted mode: a dense (cell, feature) pair number), epoch[i] = a number that changes when
 * quick_reset_db runs between records (sorted mode: the cell id; unsorted mode: constant).
 * is_new[i] = 1 when the reference counts record i as a new UMI of its (cell, gene).
 * The reset rule of quick_reset_db (:428-434) - rl_all only for features whose UMI total of the finished
 * cell is > 0 - needs the increments: incr[i] (float32, accumulated in record order as the reference does).
 * stats[0..2] += undefined reads, wild writes, overwrites.  Returns 0. */
int orl_replay(uint64_t n, const uint32_t *tree_of, const uint32_t *umi, const uint32_t *epoch, const float *incr,
               uint32_t n_trees, uint8_t *is_new, uint64_t *stats) {
    orl_tree **tr = (orl_tree **)calloc((size_t)n_trees + 1, sizeof *tr);
    float *tot = (float *)calloc((size_t)n_trees + 1, sizeof *tot);
    uint32_t *touched = (uint32_t *)malloc(((size_t)n_trees + 1) * sizeof *touched);
    uint8_t *in_list = (uint8_t *)calloc((size_t)n_trees + 1, 1);
    uint32_t n_touched = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (i && epoch[i] != epoch[i - 1]) {
            for (uint32_t k = 0; k < n_touched; k++) {
                uint32_t f = touched[k];
                if (tot[f] > 0) {
                    orl_all_out(tr[f]);
                    tot[f] = 0;
                }
                in_list[f] = 0;
            }
            n_touched = 0;
        }
        uint32_t f = tree_of[i];
        if (!in_list[f]) {
            in_list[f] = 1;
            touched[n_touched++] = f;
        }
        if (!tr[f]) {
            tr[f] = orl_new(1048576);
            orl_insert(tr[f], umi[i]);
            tot[f] += incr[i];
            is_new[i] = 1;
            continue;
        }
        if (!orl_member(tr[f], umi[i])) {
            orl_insert(tr[f], umi[i]);
            tot[f] += incr[i];
            is_new[i] = 1;
        } else {
            is_new[i] = 0;
        }
    }
    for (uint32_t f = 0; f <= n_trees; f++)
        if (tr[f]) {
            stats[0] += tr[f]->undefined_reads;
            stats[1] += tr[f]->wild_writes;
            stats[2] += tr[f]->overwrites;
            orl_free(tr[f]);
        }
    free(tr);
    free(tot);
    free(touched);
    free(in_list);
    return 0;
}
int orl_ever_written(orl_tree *t, uint64_t idx) { return idx < t->cap ? t->ever[idx] : 0; }
/* first live slot that the reference has written and that differs from other[] (another implementation's array), or -1 */
long orl_first_difference(orl_tree *t, const uint16_t *other) {
    for (uint64_t i = 0; i < t->size && i < t->cap; i++)
        if (t->ever[i] && t->node[i] != other[i]) return (long)i;
    return -1;
}
