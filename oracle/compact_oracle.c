/*
 * compact_oracle.c -- CPU restatement of the CompaCT per-slice encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (libcompact_hip.so) never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * fixtures produced by importing the reference's own src/codec in the build
 * container (oracle/gen_golden.py) and against the reference's committed golden
 * artefact data/working/testing.cct.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  The code is a sequential, literal restatement written
 * from SURVEY.md Appendix A; it shares no code with the HIP path.
 *
 * Third-party arithmetic: DEFLATE is zlib (the reference calls CPython's zlib
 * module -> system libz, src/codec/core.py:340,421; 1.2.11 in this image).  This
 * file links the same system libz.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define CCT_OK 0
#define CCT_E_MAGIC 1     /* ValueError('Image does not contain valid header'), core.py:388-389 */
#define CCT_E_ZLIB 2      /* zlib.error, core.py:421 */
#define CCT_E_OVERFLOW 3  /* OverflowError from int.to_bytes(2), core.py:506,516 */
#define CCT_E_STREAM 4    /* TypeError / IndexError on truncated or malformed token stream */
#define CCT_E_SHAPE 5     /* ValueError from numpy reshape when N % block_size != 0, core.py:245,429 */
#define CCT_E_CAP 6       /* caller's output buffer too small (oracle-only condition) */
#define CCT_E_NOMEM 7

typedef struct {
	uint32_t n_short;     /* self.info['delta'], core.py:317 */
	uint32_t n_full;      /* self.info['full'],  core.py:322 */
	uint32_t n_jump;      /* len(block_jumps),   core.py:268 */
	uint32_t n_difficult; /* len(block_deltas),  cluster.py:51-59 */
	uint32_t payload_len; /* token bytes + EOF, before DEFLATE, core.py:332 */
	uint32_t q7_violations; /* deltas outside [-2047,2048]: not representable */
} cct_oracle_stats;

/* ---------------------------------------------------------------- curve ---- */

/* Python floor division by two (curve.py:115-116 uses `//` on possibly negative ints). */
static int64_t floordiv2(int64_t a) { return (a >= 0) ? a / 2 : -((-a + 1) / 2); }
static int64_t sgn64(int64_t x) { return x < 0 ? -1 : (x > 0 ? 1 : 0); } /* curve.py:80-81 */
static int64_t abs64(int64_t x) { return x < 0 ? -x : x; }

typedef struct { int32_t *out; size_t n; int64_t width; } curve_ctx;

/* curve.py:83-138 GeneralizedHilbertCurve.generate, emitting idx((y, x)) = y*width + x
 * (curve.py:71-74, 95-96, 107-108). */
static void gilbert(curve_ctx *c, int64_t x, int64_t y, int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
	int64_t w = abs64(ax + ay);
	int64_t h = abs64(bx + by);
	int64_t dax = sgn64(ax), day = sgn64(ay);
	int64_t dbx = sgn64(bx), dby = sgn64(by);

	if (h == 1) { /* curve.py:91-101 */
		for (int64_t i = 0; i < w; i++) {
			c->out[c->n++] = (int32_t)(y * c->width + x);
			x += dax; y += day;
		}
		return;
	}
	if (w == 1) { /* curve.py:103-113 */
		for (int64_t i = 0; i < h; i++) {
			c->out[c->n++] = (int32_t)(y * c->width + x);
			x += dbx; y += dby;
		}
		return;
	}

	int64_t ax2 = floordiv2(ax), ay2 = floordiv2(ay);
	int64_t bx2 = floordiv2(bx), by2 = floordiv2(by);
	int64_t w2 = abs64(ax2 + ay2);
	int64_t h2 = abs64(bx2 + by2);

	if (2 * w > 3 * h) { /* curve.py:121-128 */
		if ((w2 % 2) && (w > 2)) { ax2 += dax; ay2 += day; }
		gilbert(c, x, y, ax2, ay2, bx, by);
		gilbert(c, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by);
	} else { /* curve.py:130-138 */
		if ((h2 % 2) && (h > 2)) { bx2 += dbx; by2 += dby; }
		gilbert(c, x, y, bx2, by2, ax2, ay2);
		gilbert(c, x + bx2, y + by2, ax, ay, bx - bx2, by - by2);
		gilbert(c, x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby),
		        -bx2, -by2, -(ax - ax2), -(ay - ay2));
	}
}

/* curve.py:45-69 generate_all / generate_and_yield with get_index=True.
 * `width` is image.shape[0], `height` is image.shape[1] (core.py:179, 235). */
int cct_oracle_curve(int width, int height, int32_t *out)
{
	curve_ctx c = { out, 0, width };
	if (width >= height) gilbert(&c, 0, 0, width, 0, 0, height);
	else                 gilbert(&c, 0, 0, 0, height, width, 0);
	return (c.n == (size_t)width * (size_t)height) ? CCT_OK : CCT_E_SHAPE;
}

/* ------------------------------------------------------- segmentation ---- */

/*
 * cluster.py:6-199 BlockPartitioner: set_delta_changes_array + initial_partition +
 * block_partition.  D = pixel values in traversal order (core.py:254-255; typed
 * values of image.flatten(), hence int32 here), O = traversal order.
 * Outputs: order_out[N] (PIXEL_ORDER) and jump_out[NB] (BLOCK_JUMPS as an array:
 * partner block index, or -1 when the block is not a key).
 */
int cct_oracle_partition(const int32_t *D, const int32_t *O, int64_t N, int bs,
                         int32_t *order_out, int32_t *jump_out, uint32_t *n_difficult)
{
	if (bs <= 0 || N % bs != 0) return CCT_E_SHAPE;
	int64_t NB = N / bs;
	uint32_t *P = (uint32_t *)calloc((size_t)N, sizeof(uint32_t));
	uint8_t *difficult = (uint8_t *)calloc((size_t)NB, 1);
	uint8_t *completed = (uint8_t *)calloc((size_t)NB, 1);
	if (!P || !difficult || !completed) { free(P); free(difficult); free(completed); return CCT_E_NOMEM; }

	/* cluster.py:33-41: prefix sum of |D[k]-D[k-1]| > 64 (the `-63 > diff` arm is dead). */
	for (int64_t k = 1; k < N; k++) {
		int64_t diff = (int64_t)D[k] - (int64_t)D[k - 1];
		if (diff < 0) diff = -diff;
		P[k] = P[k - 1] + (diff > 64 ? 1u : 0u);
	}
	/* cluster.py:51-59: difficult <=> P[end]-P[start] >= block_size/2 (float compare). */
	uint32_t nd = 0;
	for (int64_t i = 0; i < NB; i++) {
		uint32_t changes = P[i * bs + bs - 1] - P[i * bs];
		if (2.0 * (double)changes >= (double)bs) { difficult[i] = 1; nd++; }
	}
	if (n_difficult) *n_difficult = nd;
	for (int64_t i = 0; i < NB; i++) jump_out[i] = -1;

	int64_t running = 0;
	for (int64_t i = 0; i < NB; i++) { /* cluster.py:79 */
		if (!difficult[i] && !completed[i]) { /* cluster.py:89-96 */
			memcpy(order_out + running, O + i * bs, (size_t)bs * sizeof(int32_t));
			running += bs; completed[i] = 1;
			continue;
		}
		if (completed[i]) continue; /* cluster.py:98-99 */

		/* cluster.py:105-110: next_i = i+1 (the while loop never iterates);
		 * current_delta = P[next_i*bs-1] - P[start-1] in uint32 arithmetic, and
		 * Python negative indexing for i == 0 (start-1 == -1 -> P[N-1]).  (Q4) */
		uint32_t p_hi = P[(i + 1) * bs - 1];
		uint32_t p_lo = (i == 0) ? P[N - 1] : P[i * bs - 1];
		uint32_t current_delta = p_hi - p_lo;            /* wraps like numpy uint32 */
		uint32_t threshold = current_delta - 2u;         /* numpy>=2: stays uint32, wraps */

		int meshed = 0;
		for (int64_t j = 1; j <= 63 && i + j < NB; j++) { /* cluster.py:122 blocks[i+1:i+64] */
			int64_t p = i + j;
			if (completed[p]) continue; /* cluster.py:128-129 */
			/* cluster.py:138-153: interleave A,B; count D >= 65 among 2*bs-1 differences, +1 */
			const int32_t *A = D + i * bs, *B = D + p * bs;
			uint32_t up = 0;
			for (int t = 0; t < bs; t++) {
				if ((int64_t)B[t] - (int64_t)A[t] >= 65) up++;
				if (t + 1 < bs && (int64_t)A[t + 1] - (int64_t)B[t] >= 65) up++;
			}
			uint32_t num_changes = up + 1;
			if (num_changes < threshold) { /* cluster.py:158 */
				meshed = 1;
				jump_out[i] = (int32_t)p; /* cluster.py:166 */
				completed[i] = 1; completed[p] = 1;
				for (int t = 0; t < bs; t++) { /* cluster.py:173-174 */
					order_out[running + 2 * t] = O[i * bs + t];
					order_out[running + 2 * t + 1] = O[p * bs + t];
				}
				running += 2 * bs;
				break; /* first fit, cluster.py:181 */
			}
		}
		if (!meshed) { /* cluster.py:186-190 */
			memcpy(order_out + running, O + i * bs, (size_t)bs * sizeof(int32_t));
			running += bs; completed[i] = 1;
		}
	}
	free(P); free(difficult); free(completed);
	return (running == N) ? CCT_OK : CCT_E_STREAM;
}

/* ------------------------------------------------------------- encode ---- */

/* core.py:52-54 unsign(x, n_bits) = (x + 2^n) % 2^n with Python's non-negative modulo. */
static uint32_t unsign(int64_t x, int n_bits)
{
	int64_t m = (int64_t)1 << n_bits;
	int64_t r = (x + m) % m;
	if (r < 0) r += m;
	return (uint32_t)r;
}

/* zlib compressBound-safe capacity for a payload of `len` bytes plus the 13-byte header */
size_t cct_oracle_bound(int64_t N, int bs)
{
	size_t payload = (size_t)(2 * N + N / (bs > 0 ? bs : 1) + 2);
	return 13 + compressBound((uLong)payload) + 64;
}

/*
 * core.py:212-365 Encoder.encode (header at core.py:193-210).
 * img: C-order raster of 2-byte pixels, shape (width, height) = image.shape.
 * signed_seg: 1 when the caller's array dtype is int16 -- segmentation then sees
 * signed values (core.py:254 image.flatten().tolist()) while tokens always use the
 * unsigned little-endian bytes (core.py:286, 80).
 * eof: config['encoder']['end_of_file'] or -1 for None (core.py:329-330).
 * magic: 4 ASCII chars of config['magic'] (core.py:188-191).
 */
int cct_oracle_encode(const uint16_t *img, int width, int height, int bs,
                      int fractal, int seg, int deflate_on, int eof, int signed_seg,
                      const char *magic, int channels, int bytes_per_channel,
                      uint8_t *out, size_t cap, size_t *out_len, cct_oracle_stats *st)
{
	int64_t N = (int64_t)width * height;
	if (bs <= 0 || N % bs != 0) return CCT_E_SHAPE; /* core.py:245 reshape ValueError */
	int64_t NB = N / bs;
	cct_oracle_stats s; memset(&s, 0, sizeof s);
	int rc = CCT_OK;

	int32_t *O = (int32_t *)malloc((size_t)N * sizeof(int32_t));
	int32_t *order = (int32_t *)malloc((size_t)N * sizeof(int32_t));
	int32_t *blk = (int32_t *)malloc((size_t)N * sizeof(int32_t));   /* pixel_block, core.py:246-249 */
	int32_t *jump = (int32_t *)malloc((size_t)(NB ? NB : 1) * sizeof(int32_t));
	uint8_t *written = (uint8_t *)calloc((size_t)(NB ? NB : 1), 1);  /* blocks_written, core.py:244 */
	size_t pcap = (size_t)(2 * N + NB + 2);
	uint8_t *payload = (uint8_t *)malloc(pcap);
	int32_t *D = NULL;
	if (!O || !order || !blk || !jump || !written || !payload) { rc = CCT_E_NOMEM; goto done; }

	/* core.py:234-239 traversal */
	if (fractal) { rc = cct_oracle_curve(width, height, O); if (rc) goto done; }
	else for (int64_t k = 0; k < N; k++) O[k] = (int32_t)k;
	for (int64_t k = 0; k < N; k++) blk[O[k]] = (int32_t)(k / bs);
	for (int64_t b = 0; b < NB; b++) jump[b] = -1;

	if (seg) { /* core.py:251-268 */
		D = (int32_t *)malloc((size_t)N * sizeof(int32_t));
		if (!D) { rc = CCT_E_NOMEM; goto done; }
		for (int64_t k = 0; k < N; k++)
			D[k] = signed_seg ? (int32_t)(int16_t)img[O[k]] : (int32_t)img[O[k]];
		rc = cct_oracle_partition(D, O, N, bs, order, jump, &s.n_difficult);
		if (rc) goto done;
	} else {
		memcpy(order, O, (size_t)N * sizeof(int32_t));
	}

	/* core.py:274-323 token loop */
	size_t pl = 0;
	int64_t prev = 0;
	for (int64_t n = 0; n < N; n++) {
		int32_t i = order[n];
		int32_t block = blk[i];
		if (seg && jump[block] >= 0 && !written[block]) { /* core.py:290-294 */
			payload[pl++] = (uint8_t)((0x80 | (jump[block] - block)) & 0xFF);
			written[block] = 1; s.n_jump++;
		}
		int64_t cur = (int64_t)img[i];       /* unsigned LE bytes, core.py:286,297 + Pixel.update */
		int64_t delta = cur - prev;           /* core.py:313 */
		prev = cur;
		if (delta > -64 && delta < 65) {      /* core.py:316-319 */
			payload[pl++] = (uint8_t)(0x00 | unsign(delta, 7));
			s.n_short++;
		} else {                              /* core.py:322-323 */
			uint32_t v = (0xE0u << 8) | unsign(delta, 12);
			payload[pl++] = (uint8_t)((v >> 8) & 0xFF);
			payload[pl++] = (uint8_t)(v & 0xFF);
			s.n_full++;
			if (delta < -2047 || delta > 2048) s.q7_violations++;
		}
	}
	if (eof >= 0) payload[pl++] = (uint8_t)(eof & 0xFF); /* core.py:329-330 */
	s.payload_len = (uint32_t)pl;

	/* core.py:193-210 header: magic (4, big-endian int of the ASCII string), width, height
	 * (2 bytes BE each, masked to 16 bits by write_2_bytes_header), channels, bytes/channel,
	 * fractal, segmentation, deflate flags. */
	if (cap < 13) { rc = CCT_E_CAP; goto done; }
	out[0] = (uint8_t)magic[0]; out[1] = (uint8_t)magic[1]; out[2] = (uint8_t)magic[2]; out[3] = (uint8_t)magic[3];
	out[4] = (uint8_t)((width >> 8) & 0xFF);  out[5] = (uint8_t)(width & 0xFF);
	out[6] = (uint8_t)((height >> 8) & 0xFF); out[7] = (uint8_t)(height & 0xFF);
	out[8] = (uint8_t)(channels & 0xFF); out[9] = (uint8_t)(bytes_per_channel & 0xFF);
	out[10] = fractal ? 1 : 0; out[11] = seg ? 1 : 0; out[12] = deflate_on ? 1 : 0;

	if (deflate_on) { /* core.py:337-345 zlib.compress(data, level=9) */
		uLongf dl = (uLongf)(cap - 13);
		int zr = compress2(out + 13, &dl, payload, (uLong)pl, 9);
		if (zr == Z_BUF_ERROR) { rc = CCT_E_CAP; goto done; }
		if (zr != Z_OK) { rc = CCT_E_ZLIB; goto done; }
		*out_len = 13 + (size_t)dl;
	} else {
		if (cap < 13 + pl) { rc = CCT_E_CAP; goto done; }
		memcpy(out + 13, payload, pl);
		*out_len = 13 + pl;
	}
done:
	if (st) *st = s;
	free(O); free(order); free(blk); free(jump); free(written); free(payload); free(D);
	return rc;
}

/* ------------------------------------------------------------- decode ---- */

/* core.py:134-168 ByteReader with padding_len = 1: read() returns None once
 * read_pos > len - 2; peek() is unchecked (IndexError past the end). */
typedef struct { const uint8_t *b; int64_t len, pos, max_pos; } reader;
static int rd(reader *r) { if (r->pos > r->max_pos) return -1; return r->b[r->pos++]; }

/* core.py:385-402 read_header.  Returns CCT_E_MAGIC on mismatch. */
int cct_oracle_read_header(const uint8_t *file, size_t len, const char *magic,
                           int *width, int *height, int *channels, int *bpc,
                           int *fractal, int *seg, int *deflate_on)
{
	if (len < 13) return CCT_E_STREAM;
	if (memcmp(file, magic, 4) != 0) return CCT_E_MAGIC;
	*width = (file[4] << 8) | file[5];
	*height = (file[6] << 8) | file[7];
	*channels = file[8]; *bpc = file[9];
	*fractal = file[10] != 0; *seg = file[11] != 0; *deflate_on = file[12] != 0;
	return CCT_OK;
}

/*
 * core.py:404-543 Decoder.decode with out_path=None: returns the uint16 raster
 * (output.tobytes()).  `bs` and `magic` come from the decoder's own config (Q9).
 * out must hold width*height uint16 as read from the header.
 */
int cct_oracle_decode(const uint8_t *file, size_t len, int bs, const char *magic,
                      uint16_t *out, size_t out_cap_px, int *width_out, int *height_out)
{
	int W, H, ch, bpc, fractal, seg, defl;
	int rc = cct_oracle_read_header(file, len, magic, &W, &H, &ch, &bpc, &fractal, &seg, &defl);
	if (rc) return rc;
	if (width_out) *width_out = W;
	if (height_out) *height_out = H;
	int64_t N = (int64_t)W * H;
	if ((size_t)N > out_cap_px) return CCT_E_CAP;
	if (bs <= 0 || N % bs != 0) return CCT_E_SHAPE; /* core.py:429 */
	int64_t NB = N / bs;

	uint8_t *payload = NULL; size_t plen = 0;
	const uint8_t *pbytes;
	if (defl) { /* core.py:420-421 */
		size_t capz = (size_t)(2 * N + NB + 1024);
		for (;;) {
			payload = (uint8_t *)malloc(capz);
			if (!payload) return CCT_E_NOMEM;
			uLongf dl = (uLongf)capz;
			int zr = uncompress(payload, &dl, file + 13, (uLong)(len - 13));
			if (zr == Z_OK) { plen = dl; break; }
			free(payload); payload = NULL;
			if (zr == Z_BUF_ERROR && capz < ((size_t)1 << 33)) { capz *= 4; continue; }
			return CCT_E_ZLIB;
		}
		pbytes = payload;
	} else {
		pbytes = file + 13; plen = len - 13;
	}
	reader r = { pbytes, (int64_t)plen, 0, (int64_t)plen - 1 - 1 };

	int32_t *O = (int32_t *)malloc((size_t)(N ? N : 1) * sizeof(int32_t));
	int32_t *blk = (int32_t *)malloc((size_t)(N ? N : 1) * sizeof(int32_t));
	int64_t *padded = (int64_t *)malloc((size_t)(2 * N + 1) * sizeof(int64_t)); /* core.py:439-440 */
	uint8_t *completed = (uint8_t *)calloc((size_t)(N ? N : 1), 1);
	if (!O || !blk || !padded || !completed) { rc = CCT_E_NOMEM; goto done; }
	if (fractal) { rc = cct_oracle_curve(W, H, O); if (rc) goto done; }
	else for (int64_t k = 0; k < N; k++) O[k] = (int32_t)k;
	for (int64_t k = 0; k < N; k++) blk[O[k]] = (int32_t)(k / bs);
	for (int64_t k = 0; k < N; k++) { padded[2 * k] = O[k]; padded[2 * k + 1] = -1; }
	memset(out, 0, (size_t)N * sizeof(uint16_t)); /* np.zeros, core.py:436 */

	int64_t pixel = 0, prev = 0;
	for (int64_t running = 0; running < 2 * N; running++) { /* core.py:453-457 */
		if (padded[running] == -1) continue;
		int64_t index = padded[running];
		if (completed[index]) continue;
		int64_t block = blk[index];
		completed[index] = 1;

		if (r.pos >= r.len) { rc = CCT_E_STREAM; goto done; } /* peek() IndexError */
		if ((r.b[r.pos] & 0xC0) == 0x80) { /* core.py:484-494 */
			int data = rd(&r);
			if (data < 0) { rc = CCT_E_STREAM; goto done; }
			int jmp = data & 0x3F;
			int64_t bB = block + jmp;
			if (bB >= NB) { rc = CCT_E_STREAM; goto done; } /* IndexError */
			/* padded_order[running+1 : running+1+2*bs : 2] = blockB: the numpy slice is
			 * clipped at the array end and a length mismatch raises ValueError. */
			if (running + 1 + 2 * (int64_t)(bs - 1) >= 2 * N) { rc = CCT_E_STREAM; goto done; }
			for (int t = 0; t < bs; t++) padded[running + 1 + 2 * t] = O[bB * bs + t];
		}
		int data = rd(&r);
		if (data < 0) { rc = CCT_E_STREAM; goto done; } /* None & int -> TypeError */
		if ((data & 0xF0) == 0xE0) { /* core.py:500-508 */
			int b2 = rd(&r);
			if (b2 < 0) { rc = CCT_E_STREAM; goto done; }
			int64_t x = ((data << 8) | b2) & 0xFFF;
			if (x > 2048) x -= 4096;      /* signed(x, 12): x > max/2, core.py:56-60 */
			int64_t rec = prev + x;
			if (rec < 0 || rec > 65535) { rc = CCT_E_OVERFLOW; goto done; }
			pixel = rec;
		} else if ((data & 0x80) == 0x00) { /* core.py:513-516 */
			int64_t x = data & 0x7F;
			if (x > 64) x -= 128;         /* signed(x, 7) */
			int64_t rec = prev + x;
			if (rec < 0 || rec > 65535) { rc = CCT_E_OVERFLOW; goto done; }
			pixel = rec;
		} /* any other tag byte leaves `pixel` unchanged (no branch taken) */
		out[index] = (uint16_t)pixel; /* core.py:519-520 */
		prev = pixel;
	}
done:
	free(payload); free(O); free(blk); free(padded); free(completed);
	return rc;
}

int cct_oracle_abi_version(void) { return 1; }
