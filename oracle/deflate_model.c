/*
 * deflate_model.c -- CPU model of the DATA-PARALLEL restatement of zlib 1.2.11
 * `deflate(level 9, wbits 15, memLevel 8, Z_DEFAULT_STRATEGY, one-shot Z_FINISH)`.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  Third-party algorithm: zlib
 * 1.2.11 (deflate.c deflate_slow/longest_match/fill_window, trees.c), which the reference
 * reaches through CPython's zlib module (src/codec/core.py:340).  zlib is not vendored in the
 * reference; this file restates its published algorithm in the form the HIP kernels use:
 *
 *   1. hash every position (3-byte rolling hash, 15 bits) and link equal hashes (hash chains);
 *   2. for EVERY position compute what longest_match() would return with the full chain
 *      (4096 candidates) and with the shortened chain (1024, used when prev_length >= 32) --
 *      independent per position;
 *   3. lazy-match parse (deflate_slow) as a walk over "decision positions";
 *   4. blocks of 16383 symbols; per block Huffman trees (build_tree / gen_bitlen / gen_codes /
 *      scan_tree / send_tree), stored / static / dynamic choice, bit packing;
 *   5. zlib wrapper: 0x78 0xDA header, Adler-32 trailer.
 *
 * Pinned by tests/test_deflate_model.py: byte-identical to the system libz (the library the
 * oracle and the reference use) on the golden payloads and on randomised inputs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_MATCH 3
#define MAX_MATCH 258
#define WSIZE 32768
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define MAX_DIST (WSIZE - MIN_LOOKAHEAD) /* 32506 */
#define TOO_FAR 4096
#define LIT_BUFSIZE 16384
#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define END_BLOCK 256
#define MAX_BITS 15
#define MAX_BL_BITS 7

static const int extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const int extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

static uint8_t length_code[256];
static int base_length[29];
static uint8_t dist_code[512];
static int base_dist[30];
static uint16_t static_lcode[288], static_llen[288], static_dcode[30];
static int tables_ready = 0;

static unsigned bi_reverse(unsigned code, int len)
{
	unsigned res = 0;
	do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
	return res >> 1;
}

static void init_tables(void)
{ /* trees.c tr_static_init */
	if (tables_ready) return;
	int length = 0, code, n, dist = 0;
	for (code = 0; code < 28; code++) {
		base_length[code] = length;
		for (n = 0; n < (1 << extra_lbits[code]); n++) length_code[length++] = (uint8_t)code;
	}
	length_code[length - 1] = (uint8_t)code; /* length 258 -> code 28 */
	base_length[28] = 0;
	for (code = 0; code < 16; code++) {
		base_dist[code] = dist;
		for (n = 0; n < (1 << extra_dbits[code]); n++) dist_code[dist++] = (uint8_t)code;
	}
	dist >>= 7;
	for (; code < D_CODES; code++) {
		base_dist[code] = dist << 7;
		for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
	}
	uint16_t bl_count[MAX_BITS + 1] = {0};
	n = 0;
	while (n <= 143) static_llen[n++] = 8, bl_count[8]++;
	while (n <= 255) static_llen[n++] = 9, bl_count[9]++;
	while (n <= 279) static_llen[n++] = 7, bl_count[7]++;
	while (n <= 287) static_llen[n++] = 8, bl_count[8]++;
	uint16_t next_code[MAX_BITS + 1];
	unsigned c = 0;
	for (int bits = 1; bits <= MAX_BITS; bits++) { c = (c + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)c; }
	for (n = 0; n < 288; n++) static_lcode[n] = (uint16_t)bi_reverse(next_code[static_llen[n]]++, static_llen[n]);
	for (n = 0; n < D_CODES; n++) static_dcode[n] = (uint16_t)bi_reverse((unsigned)n, 5);
	tables_ready = 1;
}

static int d_code(unsigned dist) { return dist < 256 ? dist_code[dist] : dist_code[256 + (dist >> 7)]; }

/* ------------------------------------------------------------------ match finding */

typedef struct { uint16_t len4096, len1024; uint16_t dist4096, dist1024; } match_rec;

/* what longest_match(p) returns for chain limits 4096 / 1024, ignoring prev_length (the caller
 * compares with prev_length).  prevq[] = previous position with the same hash, or -1. */
static void find_matches(const uint8_t *in, int64_t L, const int32_t *prevq, int64_t p, match_rec *r)
{
	r->len4096 = r->len1024 = 0; r->dist4096 = r->dist1024 = 0;
	if (p > L - MIN_MATCH) return;  /* lookahead < MIN_MATCH: no INSERT_STRING, hash_head = NIL */
	const int64_t lookahead = L - p;
	const int max_len = lookahead < MAX_MATCH ? (int)lookahead : MAX_MATCH;  /* result is capped by lookahead */
	int best = 0; int64_t best_q = -1;
	int64_t q = prevq[p];
	int count = 0;
	/* window-slide NIL quirk (fill_window + slide_hash): near the end of the input a slide can
	 * happen exactly when strstart-relative == wsize+MAX_DIST, which turns the string at the new
	 * window origin into NIL although it is at distance MAX_DIST */
	const int64_t nil_q = (L - p < MIN_LOOKAHEAD && p >= 32506 + 32768 && (p - 32506) % 32768 == 0) ? p - 32506 : -2;
	while (q >= 0) {
		const int64_t dist = p - q;
		if (q == 0 || q == nil_q) break;                 /* NIL terminates the chain */
		if (count == 0) { if (dist > MAX_DIST) break; }   /* head: strstart - hash_head <= MAX_DIST */
		else if (dist >= MAX_DIST) break;                 /* chain: cur_match > limit */
		int len = 0;
		while (len < MAX_MATCH && p + len < L + 0 && in[q + len] == in[p + len]) len++;
		/* zlib compares up to MAX_MATCH bytes even past the lookahead, then clamps the result */
		if (len > max_len) len = max_len;
		if (len > best) { best = len; best_q = q; }
		count++;
		if (count == 1024) { r->len1024 = (uint16_t)best; r->dist1024 = (uint16_t)(best ? p - best_q : 0); }
		if (best >= max_len) break;                      /* len >= nice_match (= min(258, lookahead)) */
		if (count == 4096) break;
		q = prevq[q];
	}
	if (count < 1024) { r->len1024 = (uint16_t)best; r->dist1024 = (uint16_t)(best ? p - best_q : 0); }
	r->len4096 = (uint16_t)best; r->dist4096 = (uint16_t)(best ? p - best_q : 0);
}

/* ------------------------------------------------------------------ run rule (checker for dfl_match_run_kernel)
 * For a position whose string starts with three equal bytes the device does not walk the hash chain: it
 * derives the same record from the list of run ends.  This restates that rule on the CPU so that the not-gpu
 * tests can compare it with find_matches() on every such position.
 * spos[] = positions sorted by (hash, position), sidx[p] = index of p in spos, skey[i] = hash of spos[i]. */
static void run_rule(const uint8_t *in, int64_t L, const int32_t *spos, const uint16_t *skey, const int32_t *sidx,
                     const int32_t *re, int32_t nre, int64_t p, match_rec *out)
{
	const int64_t i = sidx[p];
	const unsigned h = skey[i];
	const int64_t lookahead = L - p;
	const int64_t max_len = lookahead < MAX_MATCH ? lookahead : MAX_MATCH;
	const uint8_t b = in[p];
	int64_t r = 3;
	while (r < max_len && in[p + r] == b) r++;
	const int has_prev = p >= 2 && in[p - 1] == b;  /* position 0 is NIL */
	int64_t best4 = has_prev ? r : 0, q4 = p - 1, best1 = best4, q1 = p - 1;
	int scan = 1;
	const int64_t nil_q = (L - p < MIN_LOOKAHEAD && p >= 32506 + 32768 && (p - 32506) % 32768 == 0) ? p - 32506 : -2;
	if (!has_prev) {
		const int have_head = i >= 1 && skey[i - 1] == h;
		const int64_t hq = have_head ? spos[i - 1] : 0;
		if (!have_head || hq == 0 || hq == nil_q || p - hq > MAX_DIST) scan = 0;
		else if (p - hq == MAX_DIST) {
			int64_t len = 0;
			while (len < max_len && in[hq + len] == in[p + len]) len++;
			best4 = best1 = len; q4 = q1 = hq; scan = 0;
		}
	}
	if (scan && best4 < max_len) {
		const int ext_ok = r < max_len;
		const uint8_t c = ext_ok ? in[p + r] : 0;
		const int64_t qw = p >= MAX_DIST ? p - MAX_DIST + 1 : 1;
		int64_t qmin4 = qw, qmin1 = qw;
		if (i >= 4096 && skey[i - 4096] == h && spos[i - 4096] > qmin4) qmin4 = spos[i - 4096];
		if (i >= 1024 && skey[i - 1024] == h && spos[i - 1024] > qmin1) qmin1 = spos[i - 1024];
		int32_t lo = 0, hi = nre;
		while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (re[mid] <= p) lo = mid + 1; else hi = mid; }
		for (int64_t t = (int64_t)lo - 1; t >= 0; t--) {
			const int64_t x = re[t];
			if (x < qmin4 + 3) break;
			if (in[x - 1] != b) continue;
			int64_t m = 0;
			while (m < r && x - 1 - m >= qmin4 && in[x - 1 - m] == b) m++;
			if (m >= 3) {
				int64_t q, len;
				if (m == r) {
					q = x - r; len = r;
					if (ext_ok && in[x] == c) { len = r + 1; while (len < max_len && in[q + len] == in[p + len]) len++; }
				} else { q = x - m; len = m; }
				if (len > best4) { best4 = len; q4 = q; }
				if (x >= qmin1 + 3) {
					const int64_t m1 = m < x - qmin1 ? m : x - qmin1;
					if (m1 == m) { if (len > best1) { best1 = len; q1 = q; } }
					else if (m1 >= 3 && m1 > best1) { best1 = m1; q1 = x - m1; }
				}
				if (best4 >= max_len) break;
			}
			if (m < r && x - 1 - m < qmin4) break;
		}
	}
	out->len4096 = (uint16_t)best4; out->len1024 = (uint16_t)best1;
	out->dist4096 = (uint16_t)(best4 ? p - q4 : 0); out->dist1024 = (uint16_t)(best1 ? p - q1 : 0);
}

/* Returns -1 when the run rule agrees with the chain walk on every run position, else the first position that
 * differs.  Lengths below MIN_MATCH are equivalent (deflate_slow ignores them). */
int64_t cct_model_check_run_rule(const uint8_t *in, size_t len)
{
	const int64_t L = (int64_t)len;
	if (L < MIN_MATCH) return -1;
	const int64_t npos = L - 2;
	int32_t *prevq = (int32_t *)malloc((size_t)(L + 1) * sizeof(int32_t));
	int32_t *head = (int32_t *)malloc(32768 * sizeof(int32_t));
	int32_t *cnt = (int32_t *)calloc(32769, sizeof(int32_t));
	int32_t *spos = (int32_t *)malloc((size_t)npos * sizeof(int32_t));
	uint16_t *skey = (uint16_t *)malloc((size_t)npos * sizeof(uint16_t));
	int32_t *sidx = (int32_t *)malloc((size_t)npos * sizeof(int32_t));
	int32_t *re = (int32_t *)malloc((size_t)(L + 1) * sizeof(int32_t));
	for (int i = 0; i < 32768; i++) head[i] = -1;
	for (int64_t p = 0; p < npos; p++) {
		const unsigned h = (((unsigned)in[p] << 10) ^ ((unsigned)in[p + 1] << 5) ^ in[p + 2]) & 0x7FFF;
		prevq[p] = head[h]; head[h] = (int32_t)p; cnt[h + 1]++;
	}
	for (int i = 0; i < 32768; i++) cnt[i + 1] += cnt[i];
	for (int64_t p = 0; p < npos; p++) {
		const unsigned h = (((unsigned)in[p] << 10) ^ ((unsigned)in[p + 1] << 5) ^ in[p + 2]) & 0x7FFF;
		const int32_t i = cnt[h]++;
		spos[i] = (int32_t)p; skey[i] = (uint16_t)h; sidx[p] = i;
	}
	int32_t nre = 0;
	for (int64_t x = 3; x < L; x++)
		if (in[x - 1] == in[x - 2] && in[x - 2] == in[x - 3] && in[x] != in[x - 1]) re[nre++] = (int32_t)x;
	int64_t bad = -1;
	for (int64_t p = 0; p < npos && bad < 0; p++) {
		if (!(in[p + 1] == in[p] && in[p + 2] == in[p])) continue;
		match_rec a, b;
		find_matches(in, L, prevq, p, &a);
		run_rule(in, L, spos, skey, sidx, re, nre, p, &b);
		const int a4 = a.len4096 >= MIN_MATCH, b4 = b.len4096 >= MIN_MATCH, a1 = a.len1024 >= MIN_MATCH, b1 = b.len1024 >= MIN_MATCH;
		if (a4 != b4 || a1 != b1) bad = p;
		else if (a4 && (a.len4096 != b.len4096 || a.dist4096 != b.dist4096)) bad = p;
		else if (a1 && (a.len1024 != b.len1024 || a.dist1024 != b.dist1024)) bad = p;
	}
	free(prevq); free(head); free(cnt); free(spos); free(skey); free(sidx); free(re);
	return bad;
}

/* ------------------------------------------------------------------ bit writer */

typedef struct { uint8_t *out; size_t pos; uint32_t bi_buf; int bi_valid; } bitw;
static void send_bits(bitw *w, unsigned value, int length)
{
	w->bi_buf |= (uint32_t)value << w->bi_valid;
	w->bi_valid += length;
	while (w->bi_valid >= 8) { w->out[w->pos++] = (uint8_t)(w->bi_buf & 0xFF); w->bi_buf >>= 8; w->bi_valid -= 8; }
}
static void bi_windup(bitw *w)
{
	if (w->bi_valid > 0) w->out[w->pos++] = (uint8_t)(w->bi_buf & 0xFF);
	w->bi_buf = 0; w->bi_valid = 0;
}

/* ------------------------------------------------------------------ Huffman trees (trees.c) */

typedef struct { uint16_t freq; uint16_t code; uint16_t dad; uint16_t len; } ct;  /* fc/dl unions split */

typedef struct {
	ct dyn_ltree[HEAP_SIZE], dyn_dtree[2 * D_CODES + 1], bl_tree[2 * BL_CODES + 1];
	uint16_t bl_count[MAX_BITS + 1];
	int heap[2 * L_CODES + 1]; int heap_len, heap_max;
	uint8_t depth[2 * L_CODES + 1];
	uint32_t opt_len, static_len;
} trees;

typedef struct { ct *tree; const uint16_t *stlen; const int *extra; int base, elems, max_length; int max_code; } tdesc;

#define SMALLER(tree, n, m, depth) (tree[n].freq < tree[m].freq || (tree[n].freq == tree[m].freq && depth[n] <= depth[m]))

static void pqdownheap(trees *s, ct *tree, int k)
{
	int v = s->heap[k], j = k << 1;
	while (j <= s->heap_len) {
		if (j < s->heap_len && SMALLER(tree, s->heap[j + 1], s->heap[j], s->depth)) j++;
		if (SMALLER(tree, v, s->heap[j], s->depth)) break;
		s->heap[k] = s->heap[j]; k = j; j <<= 1;
	}
	s->heap[k] = v;
}

static void gen_bitlen(trees *s, tdesc *d)
{
	ct *tree = d->tree; int max_code = d->max_code, h, n, m, bits, xbits, overflow = 0; uint16_t f;
	for (bits = 0; bits <= MAX_BITS; bits++) s->bl_count[bits] = 0;
	tree[s->heap[s->heap_max]].len = 0;
	for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
		n = s->heap[h];
		bits = tree[tree[n].dad].len + 1;
		if (bits > d->max_length) bits = d->max_length, overflow++;
		tree[n].len = (uint16_t)bits;
		if (n > max_code) continue;
		s->bl_count[bits]++;
		xbits = 0; if (n >= d->base) xbits = d->extra[n - d->base];
		f = tree[n].freq;
		s->opt_len += (uint32_t)f * (unsigned)(bits + xbits);
		if (d->stlen) s->static_len += (uint32_t)f * (unsigned)(d->stlen[n] + xbits);
	}
	if (overflow == 0) return;
	do {
		bits = d->max_length - 1;
		while (s->bl_count[bits] == 0) bits--;
		s->bl_count[bits]--; s->bl_count[bits + 1] += 2; s->bl_count[d->max_length]--;
		overflow -= 2;
	} while (overflow > 0);
	for (bits = d->max_length; bits != 0; bits--) {
		n = s->bl_count[bits];
		while (n != 0) {
			m = s->heap[--h];
			if (m > max_code) continue;
			if ((unsigned)tree[m].len != (unsigned)bits) {
				s->opt_len += ((uint32_t)bits - tree[m].len) * tree[m].freq;
				tree[m].len = (uint16_t)bits;
			}
			n--;
		}
	}
}

static void gen_codes(ct *tree, int max_code, const uint16_t *bl_count)
{
	uint16_t next_code[MAX_BITS + 1]; unsigned code = 0;
	for (int bits = 1; bits <= MAX_BITS; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)code; }
	for (int n = 0; n <= max_code; n++) {
		int len = tree[n].len;
		if (len == 0) continue;
		tree[n].code = (uint16_t)bi_reverse(next_code[len]++, len);
	}
}

static void build_tree(trees *s, tdesc *d)
{
	ct *tree = d->tree; int elems = d->elems, n, m, max_code = -1, node;
	s->heap_len = 0; s->heap_max = HEAP_SIZE;
	for (n = 0; n < elems; n++) {
		if (tree[n].freq != 0) { s->heap[++(s->heap_len)] = max_code = n; s->depth[n] = 0; }
		else tree[n].len = 0;
	}
	while (s->heap_len < 2) {
		node = s->heap[++(s->heap_len)] = (max_code < 2 ? ++max_code : 0);
		tree[node].freq = 1; s->depth[node] = 0; s->opt_len--;
		if (d->stlen) s->static_len -= d->stlen[node];
	}
	d->max_code = max_code;
	for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, tree, n);
	node = elems;
	do {
		n = s->heap[1]; s->heap[1] = s->heap[s->heap_len--]; pqdownheap(s, tree, 1); /* pqremove */
		m = s->heap[1];
		s->heap[--(s->heap_max)] = n; s->heap[--(s->heap_max)] = m;
		tree[node].freq = (uint16_t)(tree[n].freq + tree[m].freq);
		s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
		tree[n].dad = tree[m].dad = (uint16_t)node;
		s->heap[1] = node++;
		pqdownheap(s, tree, 1);
	} while (s->heap_len >= 2);
	s->heap[--(s->heap_max)] = s->heap[1];
	gen_bitlen(s, d);
	gen_codes(tree, max_code, s->bl_count);
}

static void scan_tree(trees *s, ct *tree, int max_code)
{
	int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
	if (nextlen == 0) max_count = 138, min_count = 3;
	tree[max_code + 1].len = (uint16_t)0xffff;
	for (n = 0; n <= max_code; n++) {
		curlen = nextlen; nextlen = tree[n + 1].len;
		if (++count < max_count && curlen == nextlen) continue;
		else if (count < min_count) s->bl_tree[curlen].freq += (uint16_t)count;
		else if (curlen != 0) { if (curlen != prevlen) s->bl_tree[curlen].freq++; s->bl_tree[16].freq++; }
		else if (count <= 10) s->bl_tree[17].freq++;
		else s->bl_tree[18].freq++;
		count = 0; prevlen = curlen;
		if (nextlen == 0) max_count = 138, min_count = 3;
		else if (curlen == nextlen) max_count = 6, min_count = 3;
		else max_count = 7, min_count = 4;
	}
}

#define SEND_CODE(w, c, tree) send_bits(w, tree[c].code, tree[c].len)

static void send_tree(trees *s, bitw *w, ct *tree, int max_code)
{
	int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
	if (nextlen == 0) max_count = 138, min_count = 3;
	for (n = 0; n <= max_code; n++) {
		curlen = nextlen; nextlen = tree[n + 1].len;
		if (++count < max_count && curlen == nextlen) continue;
		else if (count < min_count) { do { SEND_CODE(w, curlen, s->bl_tree); } while (--count != 0); }
		else if (curlen != 0) {
			if (curlen != prevlen) { SEND_CODE(w, curlen, s->bl_tree); count--; }
			SEND_CODE(w, 16, s->bl_tree); send_bits(w, (unsigned)(count - 3), 2);
		} else if (count <= 10) { SEND_CODE(w, 17, s->bl_tree); send_bits(w, (unsigned)(count - 3), 3); }
		else { SEND_CODE(w, 18, s->bl_tree); send_bits(w, (unsigned)(count - 11), 7); }
		count = 0; prevlen = curlen;
		if (nextlen == 0) max_count = 138, min_count = 3;
		else if (curlen == nextlen) max_count = 6, min_count = 3;
		else max_count = 7, min_count = 4;
	}
}

/* symbols of one block: dist == 0 -> literal lc, else match (lc = length-3, dist) */
static void compress_block(bitw *w, const uint16_t *d_buf, const uint8_t *l_buf, int nsym,
                           const ct *ltree, const ct *dtree, int use_static)
{
	for (int i = 0; i < nsym; i++) {
		unsigned dist = d_buf[i]; int lc = l_buf[i];
		if (dist == 0) {
			if (use_static) send_bits(w, static_lcode[lc], static_llen[lc]); else send_bits(w, ltree[lc].code, ltree[lc].len);
		} else {
			int code = length_code[lc];
			int sym = code + 256 + 1;
			if (use_static) send_bits(w, static_lcode[sym], static_llen[sym]); else send_bits(w, ltree[sym].code, ltree[sym].len);
			int extra = extra_lbits[code];
			if (extra) send_bits(w, (unsigned)(lc - base_length[code]), extra);
			dist--;
			code = d_code(dist);
			if (use_static) send_bits(w, static_dcode[code], 5); else send_bits(w, dtree[code].code, dtree[code].len);
			extra = extra_dbits[code];
			if (extra) send_bits(w, dist - (unsigned)base_dist[code], extra);
		}
	}
	if (use_static) send_bits(w, static_lcode[END_BLOCK], static_llen[END_BLOCK]);
	else send_bits(w, ltree[END_BLOCK].code, ltree[END_BLOCK].len);
}

/* _tr_flush_block for one block of symbols.  buf_ok: block_start >= 0 (stored block allowed). */
static void flush_block(trees *s, bitw *w, const uint16_t *d_buf, const uint8_t *l_buf, int nsym,
                        const uint8_t *stored_src, uint32_t stored_len, int buf_ok, int last)
{
	memset(s->dyn_ltree, 0, sizeof s->dyn_ltree); memset(s->dyn_dtree, 0, sizeof s->dyn_dtree);
	memset(s->bl_tree, 0, sizeof s->bl_tree);
	s->dyn_ltree[END_BLOCK].freq = 1; s->opt_len = s->static_len = 0;
	for (int i = 0; i < nsym; i++) {
		if (d_buf[i] == 0) s->dyn_ltree[l_buf[i]].freq++;
		else { s->dyn_ltree[length_code[l_buf[i]] + 256 + 1].freq++; s->dyn_dtree[d_code(d_buf[i] - 1u)].freq++; }
	}
	tdesc ld = { s->dyn_ltree, static_llen, extra_lbits, 257, L_CODES, MAX_BITS, 0 };
	static const uint16_t static_dlen[30] = {5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5};
	tdesc dd = { s->dyn_dtree, static_dlen, extra_dbits, 0, D_CODES, MAX_BITS, 0 };
	tdesc bd = { s->bl_tree, NULL, extra_blbits, 0, BL_CODES, MAX_BL_BITS, 0 };
	build_tree(s, &ld);
	build_tree(s, &dd);
	scan_tree(s, s->dyn_ltree, ld.max_code);
	scan_tree(s, s->dyn_dtree, dd.max_code);
	build_tree(s, &bd);
	int max_blindex;
	for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
		if (s->bl_tree[bl_order[max_blindex]].len != 0) break;
	s->opt_len += 3 * ((uint32_t)max_blindex + 1) + 5 + 5 + 4;
	uint32_t opt_lenb = (s->opt_len + 3 + 7) >> 3, static_lenb = (s->static_len + 3 + 7) >> 3;
	if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
	if (stored_len + 4 <= opt_lenb && buf_ok) { /* _tr_stored_block */
		send_bits(w, (0 << 1) + (unsigned)last, 3);
		bi_windup(w);
		w->out[w->pos++] = (uint8_t)(stored_len & 0xFF); w->out[w->pos++] = (uint8_t)((stored_len >> 8) & 0xFF);
		w->out[w->pos++] = (uint8_t)(~stored_len & 0xFF); w->out[w->pos++] = (uint8_t)((~stored_len >> 8) & 0xFF);
		memcpy(w->out + w->pos, stored_src, stored_len); w->pos += stored_len;
	} else if (static_lenb == opt_lenb) {
		send_bits(w, (1 << 1) + (unsigned)last, 3);
		compress_block(w, d_buf, l_buf, nsym, NULL, NULL, 1);
	} else {
		send_bits(w, (2 << 1) + (unsigned)last, 3);
		send_bits(w, (unsigned)(ld.max_code + 1 - 257), 5);
		send_bits(w, (unsigned)(dd.max_code + 1 - 1), 5);
		send_bits(w, (unsigned)(max_blindex + 1 - 4), 4);
		for (int rank = 0; rank < max_blindex + 1; rank++) send_bits(w, s->bl_tree[bl_order[rank]].len, 3);
		send_tree(s, w, s->dyn_ltree, ld.max_code);
		send_tree(s, w, s->dyn_dtree, dd.max_code);
		compress_block(w, d_buf, l_buf, nsym, s->dyn_ltree, s->dyn_dtree, 0);
	}
	if (last) bi_windup(w);
}

/* ------------------------------------------------------------------ top level */

static uint32_t adler32_model(const uint8_t *d, size_t n)
{
	uint32_t a = 1, b = 0;
	for (size_t i = 0; i < n; i++) { a = (a + d[i]) % 65521u; b = (b + a) % 65521u; }
	return (b << 16) | a;
}


size_t cct_model_deflate9(const uint8_t *in, size_t len, uint8_t *out)
{
	init_tables();
	const int64_t L = (int64_t)len;
	/* 1. hash chains */
	int32_t *prevq = (int32_t *)malloc((size_t)(L + 1) * sizeof(int32_t));
	int32_t *head = (int32_t *)malloc(32768 * sizeof(int32_t));
	for (int i = 0; i < 32768; i++) head[i] = -1;
	for (int64_t p = 0; p + MIN_MATCH <= L; p++) {
		const unsigned h = (((unsigned)in[p] << 10) ^ ((unsigned)in[p + 1] << 5) ^ in[p + 2]) & 0x7FFF;
		prevq[p] = head[h]; head[h] = (int32_t)p;
	}
	/* 2. per-position match records */
	match_rec *mr = (match_rec *)calloc((size_t)L + 2, sizeof(match_rec));
	for (int64_t p = 0; p < L; p++) find_matches(in, L, prevq, p, &mr[p]);

	/* 3. lazy parse (deflate_slow) */
	uint16_t *d_buf = (uint16_t *)malloc(((size_t)L + 2) * sizeof(uint16_t));
	uint8_t *l_buf = (uint8_t *)malloc((size_t)L + 2);
	int64_t *sym_end = (int64_t *)malloc(((size_t)L + 2) * sizeof(int64_t)); /* input position after symbol */
	int64_t *sym_top = (int64_t *)malloc(((size_t)L + 2) * sizeof(int64_t)); /* window base when tallied */
	size_t nsym = 0;
	int64_t p = 0; int avail = 0; int prev_len = 2; int prev_dist = 0;
	int64_t B = 0;       /* absolute position of window[0]: fill_window slides by wsize */
	int postloop_lit = 0;
	while (p < L) {
		{ /* fill_window(): called when lookahead < MIN_LOOKAHEAD; slides when strstart >= wsize+MAX_DIST */
			const int64_t filled_end = (L < B + 2 * WSIZE) ? L : B + 2 * WSIZE;
			if (filled_end - p < MIN_LOOKAHEAD && p - B >= WSIZE + MAX_DIST) B += WSIZE;
		}
		/* match at p given prev_length = prev_len */
		int mlen = 2, mdist = 0;
		if (prev_len < MAX_MATCH) { /* prev_length < max_lazy_match */
			const int l = (prev_len >= 32) ? mr[p].len1024 : mr[p].len4096;   /* good_match = 32 */
			const int d = (prev_len >= 32) ? mr[p].dist1024 : mr[p].dist4096;
			if (l > prev_len && l >= MIN_MATCH) { mlen = l; mdist = d; }
			else if (l >= 1 && prev_len >= MIN_MATCH) { mlen = prev_len; mdist = 0; } /* longest_match returns prev_length */
			if (mlen == MIN_MATCH && mdist > TOO_FAR) mlen = 2;                 /* new 3-byte match too far */
			/* hash_head == NIL (no candidate at all): match_length stays MIN_MATCH-1 */
			if (l == 0) mlen = 2;
		}
		if (prev_len >= MIN_MATCH && mlen <= prev_len) {
			d_buf[nsym] = (uint16_t)prev_dist; l_buf[nsym] = (uint8_t)(prev_len - MIN_MATCH);
			sym_top[nsym] = B; sym_end[nsym] = p - 1 + prev_len; nsym++;
			p = p - 1 + prev_len; avail = 0; prev_len = 2; prev_dist = 0;
		} else if (avail) {
			d_buf[nsym] = 0; l_buf[nsym] = in[p - 1]; sym_top[nsym] = B; sym_end[nsym] = p; nsym++;
			prev_len = mlen; prev_dist = mdist; p++;
		} else {
			avail = 1; prev_len = mlen; prev_dist = mdist; p++;
		}
	}
	if (avail) { d_buf[nsym] = 0; l_buf[nsym] = in[L - 1]; sym_top[nsym] = B; sym_end[nsym] = L; nsym++; postloop_lit = 1; }

	/* 4./5. blocks + wrapper */
	bitw w = { out, 0, 0, 0 };
	out[w.pos++] = 0x78; out[w.pos++] = 0xDA;
	trees *s = (trees *)malloc(sizeof(trees));
	size_t first = 0; int64_t block_start = 0;
	for (;;) {
		size_t n = nsym - first;
		int last = 1;
		/* flush when last_lit == lit_bufsize-1; the literal tallied AFTER the main loop never flushes */
		if (n >= LIT_BUFSIZE - 1 && !(n == LIT_BUFSIZE - 1 && postloop_lit)) { n = LIT_BUFSIZE - 1; last = 0; }
		int64_t end, base;
		if (last) { end = L; base = B; }
		else { end = sym_end[first + n - 1]; base = sym_top[first + n - 1]; }
		flush_block(s, &w, d_buf + first, l_buf + first, (int)n, in + block_start, (uint32_t)(end - block_start),
		            block_start >= base, last);
		block_start = end; first += n;
		if (last) break;
	}
	const uint32_t ad = adler32_model(in, len);
	out[w.pos++] = (uint8_t)(ad >> 24); out[w.pos++] = (uint8_t)(ad >> 16); out[w.pos++] = (uint8_t)(ad >> 8); out[w.pos++] = (uint8_t)ad;
	free(prevq); free(head); free(mr); free(d_buf); free(l_buf); free(sym_end); free(sym_top); free(s);
	return w.pos;
}
