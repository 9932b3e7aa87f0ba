// Host side of libcompact_hip.so: device context, traversal-table cache, workspaces, the
// DEFLATE/INFLATE stage (system libz on a host thread team -- the same library the reference
// reaches through CPython's zlib module, core.py:340,421) and the extern "C" entry points
// declared in include/compact_hip.h.
#include <dlfcn.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/compact_hip.h"
#include "cct_internal.h"
#include "host.h"

namespace cct {
bool gilbert_table(int width, int height, int32_t *out);

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

std::mutex g_mu;

namespace {

struct ShapeTables {  // everything that depends on (width, height) only; lives in HBM
	int32_t *d_lut = nullptr;  // traversal order O[N]
	bool tiled = false;        // traversal = aligned 64x64 tiles -> encode_tiles_kernel applies
	int n_tiles = 0, n_orient = 0;
	uint32_t *d_org = nullptr;
	uint8_t *d_orient = nullptr;
	uint16_t *d_pat = nullptr;
	// staged pipeline (encode_pipe.hip): every tile is a 16x16 grid of 4x4-pixel traversal blocks
	bool pipe = false;
	uint32_t *d_ptab = nullptr;       // n_orient * 128 * 4
	uint32_t *d_ptab2 = nullptr;      // n_orient * 2 * 64 * 4
	uint32_t *d_btab = nullptr;       // n_orient * 256
	uint32_t *d_otab = nullptr;       // 4 * 16
	uint32_t *d_ttab = nullptr;       // 16 * 4
	uint32_t *d_htab = nullptr;       // n_orient * 32 * 2 (encode_stream.hip: the look-ahead quadrant of a tile)
	PipeTiles tiles;                  // host copy: travels in the kernel arguments
};

// Everything one encode batch in flight owns: stream, workspaces, the captured DEFLATE graph.  There are two of them.
// Several kernels of the DEFLATE pass are latency-bound and leave most of the chip idle (the Huffman tree kernel runs one
// serial heap replay per block for over a millisecond; the match kernel waits on scattered loads), so a second batch on a
// stream of its own fills the gaps: two callers get two slots and their kernel chains interleave on the device.
struct EncSlot {
	std::mutex *mu = nullptr;      // slot 0: g_mu (it shares the main stream with the plumbing calls), slot 1: its own
	hipStream_t stream = nullptr;  // slot 0: the main stream
	// the ~25 launches of one DEFLATE pass, captured once per argument set and replayed as a graph: fewer host
	// calls and no dispatch gaps between the kernels when a decode shares the queue processor
	struct ZGraph { std::vector<uint8_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t last_use = 0; };
	std::vector<ZGraph> z_graphs;  // a batch deflated in several passes has one argument set per pass (round 2 kept ONE graph and
	uint64_t z_clock = 0;          // captured it again for every pass of such a batch: 5 ms per step at 512 x 1024^2)
	// the memset + four launches of the transform+pack pipeline, likewise (a few argument sets: callers rotate batches)
	struct PipeGraph { std::vector<uint8_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t last_use = 0; };
	std::vector<PipeGraph> p_graphs;
	uint64_t p_clock = 0;
	// packed archives leave the device on their own stream from one of two buffers, after the slot has been
	// released: the next encode call may start its kernels while this one's files are still on the wire
	hipStream_t stream_copy = nullptr;
	hipEvent_t ev_pack[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
	DevBuf z_packed2[2];
	unsigned pack_slot = 0;
	// encode workspaces
	DevBuf e_role, e_lidx, e_lmask, e_lcur, e_images, e_payload, e_sizes, e_status, e_stats;
	DevBuf e_toff, e_pairrec, e_spill, e_tflag;  // staged pipeline: tile offsets, meshed-pair records, difficult-list spill
	DevBuf e_hand;                               // streaming kernel: hand-off words and tickets
	DevBuf h_stage;  // pinned host staging (payloads)
	DevBuf h_small[2];  // pinned landing place of a call's sizes / status / statistics (a copy to pageable memory blocks the host until it has happened)
	// device DEFLATE workspaces
	DevBuf z_vals_in, z_vals_out, z_mr, z_rec, z_exitp, z_exitc, z_sym, z_bentry, z_bsym,
	    z_small, z_bend, z_meta, z_tables, z_sorttmp, z_out, z_outsizes, z_in, z_insizes, z_packed, z_packoffs, z_gen, z_runs;
	// match records are valid by tag (deflate_kernels.hip MatchRec): the device counter z_gen has run z_gen_passes times since
	// the buffer z_mr_cleared (z_mr at that time) was last zeroed together with it
	const void *z_mr_cleared = nullptr;
	size_t z_mr_cleared_cap = 0;
	unsigned z_gen_passes = 0;
	// timing events of an encode call, two sets: with the slot handed on before the host has synchronised (encode_batch_impl,
	// "queue ahead") the next call records its own while this one has not read its durations yet
	hipEvent_t ev_k0s[2] = {nullptr, nullptr}, ev_k1s[2] = {nullptr, nullptr};  // around the transform+pack stage
	hipEvent_t ev_z0s[2] = {nullptr, nullptr}, ev_z1s[2] = {nullptr, nullptr};  // around the device DEFLATE pass
	hipEvent_t ev_small[2] = {nullptr, nullptr};                                // sizes and status have reached the host
	hipStream_t stream_side = nullptr;                                          // side branch of the DEFLATE pass (launch_deflate)
	hipEvent_t ev_fork[4] = {nullptr, nullptr, nullptr, nullptr};
	unsigned call_parity = 0;
	DevBuf *all_bufs[48];
	int n_bufs = 0;
	EncSlot()
	{
		DevBuf *b[] = {&e_role, &e_lidx, &e_lmask, &e_lcur, &e_images, &e_payload, &e_sizes, &e_status, &e_stats, &e_toff, &e_pairrec,
		               &e_spill, &e_tflag, &e_hand, &h_stage, &z_vals_in, &z_vals_out, &z_mr, &z_rec, &z_exitp, &z_exitc,
		               &z_sym, &z_bentry, &z_bsym, &z_small, &z_bend, &z_meta, &z_tables, &z_sorttmp, &z_out, &z_outsizes, &z_in,
		               &z_insizes, &z_packed, &z_packoffs, &z_packed2[0], &z_packed2[1], &z_gen, &z_runs, &h_small[0], &h_small[1]};
		for (DevBuf *p : b) all_bufs[n_bufs++] = p;
		h_stage.pinned_host = true;
		h_small[0].pinned_host = h_small[1].pinned_host = true;
	}
	EncSlot(const EncSlot &) = delete;
	EncSlot &operator=(const EncSlot &) = delete;
};
constexpr int N_ENC_SLOTS = 2;

// The same for a decode batch in flight (cct_decode_batch, cct_zlib_decompress_batch): stream, workspaces, events.  With two of
// them the archive upload of one batch runs under the INFLATE and decode kernels of the other (what bounds a decode-only
// caller: BASELINE configs[4]).
struct DecSlot {
	hipStream_t stream = nullptr;  // slot 0: Context::stream_dec
	DevBuf d_role, d_slot, d_jord, d_jval, d_payload, d_sizes, d_status, d_images, d_pcache;
	DevBuf d_arch, d_archoffs, d_zstatus;
	DevBuf dh_stage;  // pinned host staging of inflated payloads (host INFLATE path)
	hipEvent_t ev_d0 = nullptr, ev_d1 = nullptr, ev_k_dec0 = nullptr, ev_k_dec1 = nullptr;
	hipEvent_t ev_ws = nullptr;  // recorded after a launch on ANOTHER stream that uses this slot's workspaces (cct_decode_payload_dev)
	DevBuf *all_bufs[16];
	int n_bufs = 0;
	DecSlot()
	{
		DevBuf *b[] = {&d_role, &d_slot, &d_jord, &d_jval, &d_payload, &d_sizes, &d_status, &d_images, &d_pcache, &d_arch, &d_archoffs,
		               &d_zstatus, &dh_stage};
		for (DevBuf *p : b) all_bufs[n_bufs++] = p;
		dh_stage.pinned_host = true;
	}
	DecSlot(const DecSlot &) = delete;
	DecSlot &operator=(const DecSlot &) = delete;
};
constexpr int N_DEC_SLOTS = 2;

struct Context {
	bool ready = false;
	pid_t pid = 0;
	int device = -1;
	hipStream_t stream = nullptr;  // main stream: plumbing calls, cct_encode_payload_dev, encode slot 0
	int use_graph = 1;
	int deflate_fork = 1;   // independent kernels of the DEFLATE pass on a side branch (launch_deflate)
	int queue_ahead = 1;    // an encode call hands its slot on before it waits for the sizes (encode_batch_impl)
	int decode_yields = 1;  // decode kernels issued next to an encode wait for the next transform+pack stage to end (cct_decode_batch)
	int compact_recs = 1;  // sort records that carry the first five string bytes (slices below 4 MiB; deflate_kernels.hip "Sort records")
	int enc_slots = 1;  // option "encode_slots": encode batches on the device at a time.  Default 1: on two of the three boxes
	                    // measured a second batch in flight cost more (each kernel slows down next to another batch's
	                    // DEFLATE pass) than it filled (bench.py reports both settings: stages.other_encode_slot_setting)
	hipStream_t stream_dec = nullptr;  // decode runs on its own stream so it can overlap an encode in flight
	std::map<std::pair<int, int>, ShapeTables> luts;  // (width,height) -> device tables
	int use_tiles = 1;  // option "tile_path": 1 default choice among the tile paths (the streaming kernel wherever it applies), 4 the
	                    // streaming kernel (encode_stream.hip), 3 the four-kernel pipeline (encode_pipe.hip), 2 the
	                    // one-workgroup-per-slice tile kernel, 0 the generic LUT-gather kernel
	int stream_tpg = STREAM_TPG;  // tuning option "stream_tpg": tiles per workgroup of the streaming kernel (1, 2, 4)
	int pipe_tpw = 0, pipe_timing = 0;  // tuning options "pipe_tpw", "pipe_timing" (then "pipe_us_k1/k2/k3" hold the last kernel times)
	float pipe_us[4] = {0, 0, 0, 0};
	int last_path = -1; // read-only option "last_encode_path": which stage (i) implementation the last encode used (0 generic, 1 pipeline, 2 tile kernel, 3 streaming kernel)
	int dbg_skip = 0;   // option "debug_skip": phase-ablation mask for tuning runs (outputs invalid when set)
	int device_deflate = 1;  // option "device_deflate": 0 = DEFLATE stage on the host thread team (libz)
	int device_inflate = 1;  // option "device_inflate": 1 = INFLATE on the device (inflate_kernels.hip, speculative lane-parallel
	                         // decode), 0 = libz on the host thread team (same bytes; bounded by the host CPUs the process may use)
	int zlib_threads = 0;
	int wg_threads = 1024;
	int dec_slots = N_DEC_SLOTS;  // option "decode_slots"
	int inflate_lanes = 0;  // option "inflate_lanes": 256 / 512 lanes per stream in the INFLATE kernel, 0 = 512 unless an encode call is
	                        // in flight (next to an encode batch the narrower workgroup is the faster one, see inflate_kernels.hip)
	int last_inflate_lanes = 0;  // read-only option "last_inflate_lanes"
};
std::atomic<int> g_encodes_in_flight{0};
// gate between the decode and the encode stream (sched_kernels.hip): device counters and what has been issued so far
uint32_t *g_gate = nullptr;
std::atomic<uint32_t> g_gate_passes_issued{0};
struct EncodeInFlight {
	EncodeInFlight() { g_encodes_in_flight.fetch_add(1, std::memory_order_relaxed); }
	~EncodeInFlight() { g_encodes_in_flight.fetch_sub(1, std::memory_order_relaxed); }
};

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

Context g_ctx;
}  // namespace
QuiesceLock g_quiesce;
thread_local int tl_api_depth = 0;
namespace {
int inflate_lanes_now()
{
	const int lanes = g_ctx.inflate_lanes ? g_ctx.inflate_lanes : (g_encodes_in_flight.load(std::memory_order_relaxed) > 0 ? 256 : 512);
	g_ctx.last_inflate_lanes = lanes;
	return lanes;
}
EncSlot *g_enc = new EncSlot[N_ENC_SLOTS];
DecSlot *g_dec = new DecSlot[N_DEC_SLOTS];
void reset_ctx()
{
	g_ctx = Context();
	delete[] g_enc; g_enc = new EncSlot[N_ENC_SLOTS];
	delete[] g_dec; g_dec = new DecSlot[N_DEC_SLOTS];
}
// timings of the calling thread's most recent batch call (cct_last_timings): per thread, so that an encode and a
// decode driven from two threads do not overwrite each other's numbers
thread_local float tl_enc_kernel_ms = 0, tl_dec_kernel_ms = 0, tl_d2h_ms = 0, tl_deflate_ms = 0, tl_inflate_ms = 0, tl_h2d_ms = 0;
std::mutex g_mu1;     // encode slot 1
// A call that has handed its slot on before its files left the device ("queue ahead", encode_batch_impl) still owns one of the
// slot's two packed-archive buffers until it has queued the copy out of it; the call after next waits here before packing into it.
std::mutex g_pack_mu;
std::condition_variable g_pack_cv;
bool g_pack_busy[N_ENC_SLOTS][2] = {};
std::mutex g_mu_dec[N_DEC_SLOTS];  // the decode slots; like g_mu / g_mu1 they outlive reset_ctx(), which replaces the slot objects
std::mutex g_mu_lut;  // the per-shape table cache (taken after g_mu by encode, alone by decode)

int default_device()
{
	const char *lr = getenv("LOCAL_RANK");
	return lr ? atoi(lr) : 0;
}

// Asynchronous copies that target locals (std::vector on the stack frame) or caller memory must have landed before an
// error return unwinds the frame: declare one of these AFTER those locals; it drains the stream unless disarmed.
struct DrainOnExit {
	hipStream_t s; bool armed = true;
	explicit DrainOnExit(hipStream_t st) : s(st) {}
	~DrainOnExit() { if (armed) (void)hipStreamSynchronize(s); }
	void disarm() { armed = false; }
};

}  // namespace

int ensure_ctx(int device)
{
	if (g_ctx.ready && g_ctx.pid != getpid()) {
		// A child forked AFTER the parent initialised the device inherits a HIP runtime it cannot use (ROCm does not
		// support that state, and re-creating the context on top of it is not safe either).  Callers that fan work out over
		// processes (scripts/evaluate.py:107) must fork before the first device call: each child then initialises its own.
		return fail(CCT_E_DEVICE, "this process was forked after the library had initialised the GPU in its parent (pid %d): "
		                          "fork workers before the first cct_* device call", (int)g_ctx.pid);
	}
	if (g_ctx.ready) {
		if (device >= 0 && device != g_ctx.device)
			return fail(CCT_E_ARG, "library already bound to device %d", g_ctx.device);
		HIP_TRY(hipSetDevice(g_ctx.device));
		return CCT_OK;
	}
	// The runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  The library keeps up to eight
	// streams busy (two encode slots with copy and side streams, two decode slots, the size gather), and two streams on one queue
	// run one after the other: with a stream more than before, the packed-archive copy and the next DEFLATE pass shared a queue
	// and a 1024^2 pass started 3 ms late (BASELINE configs[3]: 14.9 K instead of 17 K MPixels/s); with another assignment the
	// side branch of a pass shared one with the decode (configs[1]: DEFLATE 4.1 instead of 3.6 ms).  Eight queues, unless the
	// user has set the variable; it is read when the runtime initialises, i.e. by the first HIP call of the process below.
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(CCT_E_DEVICE, "no HIP device available (%s); libcompact_hip has no CPU fallback",
		            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	const int dev = device >= 0 ? device : default_device();
	if (dev >= count) return fail(CCT_E_DEVICE, "device %d requested but only %d visible", dev, count);
	HIP_TRY(hipSetDevice(dev));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(CCT_E_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", dev, prop.gcnArchName);
	HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream_dec, hipStreamNonBlocking));
	for (int k = 0; k < N_ENC_SLOTS; k++) {
		EncSlot &E = g_enc[k];
		E.mu = k == 0 ? &g_mu : &g_mu1;
		if (k == 0) E.stream = g_ctx.stream;
		else HIP_TRY(hipStreamCreateWithFlags(&E.stream, hipStreamNonBlocking));
		for (int q = 0; q < 2; q++) {
			HIP_TRY(hipEventCreate(&E.ev_k0s[q]));
			HIP_TRY(hipEventCreate(&E.ev_k1s[q]));
			HIP_TRY(hipEventCreate(&E.ev_z0s[q]));
			HIP_TRY(hipEventCreate(&E.ev_z1s[q]));
			HIP_TRY(hipEventCreateWithFlags(&E.ev_small[q], hipEventDisableTiming));
		}
		for (int q = 0; q < 4; q++) HIP_TRY(hipEventCreateWithFlags(&E.ev_fork[q], hipEventDisableTiming));
	}
	for (int k = 0; k < N_DEC_SLOTS; k++) {
		DecSlot &D = g_dec[k];
		if (k == 0) D.stream = g_ctx.stream_dec;
		else HIP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreate(&D.ev_d0));
		HIP_TRY(hipEventCreate(&D.ev_d1));
		HIP_TRY(hipEventCreate(&D.ev_k_dec0));
		HIP_TRY(hipEventCreate(&D.ev_k_dec1));
		HIP_TRY(hipEventCreateWithFlags(&D.ev_ws, hipEventDisableTiming));
	}
	HIP_TRY(deflate_init_tables());  // __constant__ tables of the DEFLATE kernels, shared by both encode slots
	HIP_TRY(hipMalloc(&g_gate, 256));
	HIP_TRY(hipMemset(g_gate, 0, 256));
	g_gate_passes_issued = 0;
	g_ctx.device = dev;
	g_ctx.pid = getpid();
	if (g_ctx.zlib_threads <= 0) {
		// host team size: the CPUs this process may actually use.  Under a cgroup CPU quota (cpu.max) more
		// runnable threads than about twice the quota only get the whole process throttled.
		unsigned hc = std::thread::hardware_concurrency();
		int nt = hc ? (int)hc : 1;
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			long long quota = 0, period = 0;
			if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
				nt = std::min(nt, (int)std::max<long long>(1, 2 * ((quota + period - 1) / period)));
			fclose(f);
		}
		g_ctx.zlib_threads = nt;
	}
	g_ctx.ready = true;
	return CCT_OK;
}

hipStream_t main_stream() { return g_ctx.stream; }
int bound_device() { return g_ctx.device; }
bool forked_after_init() { return g_ctx.ready && g_ctx.pid != getpid(); }

namespace {

// Tables of the staged pipeline.  Inside every 64x64 tile the traversal must walk aligned 4x4-pixel blocks (16
// positions each), and every block must be walked quadrant by quadrant (2x2 pixels, 4 positions each) with quarter 0
// = top-left or bottom-right quadrant, quarter 2 = the other one, quarter 1 = bottom-left or top-right, quarter 3 = the
// other one.  The generalized Hilbert curve on power-of-two squares does; anything else keeps the older kernels.
void build_pipe_tables(const std::vector<int32_t> &O, int width, const std::vector<uint32_t> &org,
                       const std::vector<uint8_t> &orient, ShapeTables &t)
{
	t.pipe = false;
	const int nt = (int)org.size();
	const int no = t.n_orient;
	std::vector<uint16_t> rtab((size_t)no * 256, 0xFFFF);
	std::vector<uint32_t> tile_last(no, 0), tile_mid(no, 0);
	std::vector<std::vector<int>> bpat;  // block orientations: raster index (row*4+col) of the 16 positions
	for (int ti = 0; ti < nt; ti++) {
		const int32_t *k = O.data() + (size_t)ti * 4096;
		const int to = orient[ti];
		for (int b = 0; b < 256; b++) {
			int lo = k[b * 16];
			for (int i = 1; i < 16; i++) lo = std::min(lo, k[b * 16 + i]);
			const int d0 = lo - (int)org[ti];
			const int by4 = d0 / width, bx4 = d0 % width;
			if (by4 % 4 || bx4 % 4 || by4 >= 64 || bx4 >= 64) return;
			std::vector<int> pat(16);
			for (int i = 0; i < 16; i++) {
				const int d = k[b * 16 + i] - lo, dy = d / width, dx = d % width;
				if (dx >= 4 || dy >= 4) return;
				pat[i] = dy * 4 + dx;
			}
			int bo = -1;
			for (size_t q = 0; q < bpat.size(); q++) if (bpat[q] == pat) { bo = (int)q; break; }
			if (bo < 0) { if (bpat.size() == 4) return; bpat.push_back(pat); bo = (int)bpat.size() - 1; }
			const uint16_t ent = (uint16_t)(b | (bo << 8));
			uint16_t &slot = rtab[(size_t)to * 256 + (by4 / 4) * 16 + bx4 / 4];
			if (slot != 0xFFFF && slot != ent) return;  // tiles of one orientation must agree
			slot = ent;
		}
		tile_last[to] = (uint32_t)(k[4095] - (int)org[ti]);
		tile_mid[to] = (uint32_t)(k[2047] - (int)org[ti]);
	}
	for (uint16_t e : rtab) if (e == 0xFFFF) return;
	// per-lane entries of a tile workgroup and the block table of the mask kernel
	std::vector<uint32_t> ptab((size_t)no * 128 * 4, 0), btab((size_t)no * 256, 0);
	{
		std::vector<int> done(no, 0);
		for (int ti = 0; ti < nt; ti++) {
			const int to = orient[ti];
			if (done[to]) continue;
			done[to] = 1;
			const int32_t *k = O.data() + (size_t)ti * 4096;
			for (int lane = 0; lane < 128; lane++) {
				const int by = lane >> 3, bxp = lane & 7;
				uint32_t *e = ptab.data() + ((size_t)to * 128 + lane) * 4;
				e[0] = (uint32_t)rtab[(size_t)to * 256 + by * 16 + 2 * bxp] | (uint32_t)rtab[(size_t)to * 256 + by * 16 + 2 * bxp + 1] << 16;
				for (int h = 0; h < 2; h++) {
					const int kb = (int)((e[0] >> (16 * h)) & 0xFF);
					e[1 + h] = kb == 0 ? 0xFFFFFFFFu : (uint32_t)(k[kb * 16 - 1] - (int)org[ti]);
				}
			}
			for (int b = 0; b < 256; b++) {
				int lo = k[b * 16];
				for (int i = 1; i < 16; i++) lo = std::min(lo, k[b * 16 + i]);
				uint32_t bo = 0;
				for (int r = 0; r < 256; r++) if ((rtab[(size_t)to * 256 + r] & 0xFF) == b) bo = rtab[(size_t)to * 256 + r] >> 8;
				btab[(size_t)to * 256 + b] = (uint32_t)(lo - (int)org[ti]) | bo << 24;
			}
		}
	}
	// half-tile waves: the 64 block pairs whose blocks lie in the first / second 128 traversal blocks, raster order
	std::vector<uint32_t> ptab2((size_t)no * 2 * 64 * 4, 0);
	for (int to = 0; to < no; to++)
		for (int half = 0; half < 2; half++) {
			int cnt = 0;
			for (int lane = 0; lane < 128; lane++) {
				const uint32_t *e = ptab.data() + ((size_t)to * 128 + lane) * 4;
				const int ka = (int)(e[0] & 0xFF), kb = (int)((e[0] >> 16) & 0xFF);
				if ((ka >> 7) != (kb >> 7)) return;  // a pair of blocks straddles the halves: not the structure assumed
				if ((ka >> 7) != half) continue;
				if (cnt == 64) return;
				uint32_t *d = ptab2.data() + (((size_t)to * 2 + half) * 64 + cnt) * 4;
				d[0] = e[0]; d[1] = e[1]; d[2] = e[2];
				d[3] = (uint32_t)((lane >> 3) * 4 * width + (lane & 7) * 8);
				cnt++;
			}
			if (cnt != 64) return;
			// the half must be a 64x32 (block rows r0..r0+7) or a 32x64 (block-pair columns c0..c0+3) rectangle
			const uint32_t *d0 = ptab2.data() + ((size_t)to * 2 + half) * 64 * 4;
			uint32_t geom = 0xFFFFFFFFu;
			for (int vertical = 0; vertical < 2 && geom == 0xFFFFFFFFu; vertical++)
				for (int first = 0; first < 16 && geom == 0xFFFFFFFFu; first += vertical ? 4 : 8) {
					bool ok = true;
					for (int l = 0; l < 64 && ok; l++) {
						const uint32_t want = vertical ? (uint32_t)((l >> 2) * 4 * width + (first + (l & 3)) * 8)
						                               : (uint32_t)((first + (l >> 3)) * 4 * width + (l & 7) * 8);
						ok = d0[l * 4 + 3] == want;
					}
					if (ok) geom = (uint32_t)vertical | (uint32_t)first << 8;
				}
			if (geom == 0xFFFFFFFFu) return;
			t.tiles.geom[to * 2 + half] = geom;
		}
	if (nt > 256) return;
	for (int ti = 0; ti < nt; ti++) {
		if (org[ti] >= (1u << 24)) return;
		t.tiles.orgo[ti] = org[ti] | ((uint32_t)orient[ti] << 24);
	}
	for (int to = 0; to < no; to++) { t.tiles.last[to] = tile_last[to]; t.tiles.mid[to] = tile_mid[to]; }
	std::vector<uint32_t> otab(64, 0);
	for (size_t bo = 0; bo < bpat.size(); bo++) {
		const std::vector<int> &pat = bpat[bo];
		int quad[4];
		for (int q = 0; q < 4; q++) {
			const int r = pat[4 * q] / 4, c = pat[4 * q] % 4;
			quad[q] = (r >> 1) * 2 + (c >> 1);  // 0 TL, 1 TR, 2 BL, 3 BR
			for (int i = 1; i < 4; i++)
				if (((pat[4 * q + i] / 4) >> 1) * 2 + ((pat[4 * q + i] % 4) >> 1) != quad[q]) return;
		}
		if (!((quad[0] == 0 && quad[2] == 3) || (quad[0] == 3 && quad[2] == 0))) return;
		if (!((quad[1] == 2 && quad[3] == 1) || (quad[1] == 1 && quad[3] == 2))) return;
		uint32_t *ot = otab.data() + bo * 16;
		for (int j = 0; j < 8; j++) {
			uint32_t sel = 0;
			for (int h = 0; h < 2; h++) {
				const int r = pat[2 * j + h] / 4, c = pat[2 * j + h] % 4;
				const uint32_t byte0 = (uint32_t)((r & 1) * 4 + (c & 1) * 2);  // v_perm source: top dword bytes 0-3, bottom 4-7
				sel |= (byte0 | ((byte0 + 1) << 8)) << (16 * h);
			}
			ot[j] = sel;
		}
		ot[8] = (quad[0] == 3 ? 1u : 0u) | (quad[1] == 1 ? 2u : 0u);
	}
	// token bytes of a 4-pixel group by its two-byte mask: v_perm selectors over {X: bytes 4-7, P: bytes 0-3}
	std::vector<uint32_t> ttab(64, 0);
	for (int f = 0; f < 16; f++) {
		uint8_t seq[8];
		int n = 0;
		for (int px = 0; px < 4; px++) { if (f >> px & 1) seq[n++] = (uint8_t)(4 + px); seq[n++] = (uint8_t)px; }
		for (int i = n; i < 8; i++) seq[i] = 0x0C;  // constant 0x00
		ttab[f * 4 + 0] = seq[0] | seq[1] << 8 | seq[2] << 16 | (uint32_t)seq[3] << 24;
		ttab[f * 4 + 1] = seq[4] | seq[5] << 8 | seq[6] << 16 | (uint32_t)seq[7] << 24;
		ttab[f * 4 + 2] = (uint32_t)n;
		uint32_t keep = 0x7F7F7F7Fu;  // bits of the deltas' low bytes that reach the stream: 7 of a short token, 8 of a full token's second byte
		for (int px = 0; px < 4; px++) if (f >> px & 1) keep |= 0x80u << (8 * px);
		ttab[f * 4 + 3] = keep;
	}
	// the first 64 traversal blocks of a tile (the look-ahead of the previous tile's mesh search): a 32x32-pixel quadrant, 8
	// block rows of 4 block pairs.  Entry (row * 4 + pair) of htab is the pair's ptab word; the quadrant's origin inside the
	// tile travels in the kernel arguments (tiles.qorg), so the rows of the look-ahead are requested without a table lookup
	std::vector<uint32_t> htab((size_t)no * 32 * 2, 0);
	for (int to = 0; to < no; to++) {
		int cnt = 0, r0 = 16, c0 = 8;
		for (int lane = 0; lane < 128; lane++) {
			const uint32_t *e = ptab.data() + ((size_t)to * 128 + lane) * 4;
			const int ka = (int)(e[0] & 0xFF), kb = (int)((e[0] >> 16) & 0xFF);
			if ((ka < 64) != (kb < 64)) return;
			if (ka >= 64) continue;
			r0 = std::min(r0, lane >> 3); c0 = std::min(c0, lane & 7);
			cnt++;
		}
		if (cnt != 32 || r0 > 8 || c0 > 4) return;
		for (int lane = 0; lane < 128; lane++) {
			const uint32_t *e = ptab.data() + ((size_t)to * 128 + lane) * 4;
			if ((int)(e[0] & 0xFF) >= 64) continue;
			const int r = (lane >> 3) - r0, c = (lane & 7) - c0;
			if (r < 0 || r >= 8 || c < 0 || c >= 4) return;  // not an 8 x 4 rectangle of block pairs
			htab[((size_t)to * 32 + r * 4 + c) * 2] = e[0];
			htab[((size_t)to * 32 + r * 4 + c) * 2 + 1] = (uint32_t)((lane >> 3) * 4 * width + (lane & 7) * 8);
		}
		t.tiles.qorg[to] = (uint32_t)(r0 * 4 * width + c0 * 8);
	}
	if (hipMalloc(&t.d_ptab, ptab.size() * 4) != hipSuccess) return;
	if (hipMalloc(&t.d_btab, btab.size() * 4) != hipSuccess) return;
	if (hipMalloc(&t.d_ptab2, ptab2.size() * 4) != hipSuccess) return;
	if (hipMemcpy(t.d_ptab2, ptab2.data(), ptab2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMalloc(&t.d_htab, htab.size() * 4) != hipSuccess) return;
	if (hipMemcpy(t.d_htab, htab.data(), htab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMalloc(&t.d_otab, otab.size() * 4) != hipSuccess) return;
	if (hipMalloc(&t.d_ttab, ttab.size() * 4) != hipSuccess) return;
	if (hipMemcpy(t.d_ptab, ptab.data(), ptab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMemcpy(t.d_btab, btab.data(), btab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMemcpy(t.d_otab, otab.data(), otab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMemcpy(t.d_ttab, ttab.data(), ttab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return;
	t.pipe = true;
}

// Does the traversal decompose into aligned 64x64 tiles (4096 consecutive positions each)?  If so
// build what encode_tiles_kernel needs: per tile its origin and which pattern it follows, and per
// distinct pattern the LDS byte offset (row-XOR-swizzled raster image) of every position.
void build_tile_tables(const std::vector<int32_t> &O, int width, ShapeTables &t)
{
	t.tiled = false;
	const size_t N = O.size();
	if (N == 0 || N % 8192 != 0 || width % 8 != 0) return;
	const int nt = (int)(N / 4096);
	if (nt > TILE_MAX_TILES) return;
	std::vector<uint32_t> org(nt);
	std::vector<uint8_t> orient(nt);
	std::vector<std::vector<uint16_t>> pats;
	std::vector<uint16_t> cur(4096);
	for (int ti = 0; ti < nt; ti++) {
		const int32_t *k = O.data() + (size_t)ti * 4096;
		int32_t lo = k[0];
		for (int i = 1; i < 4096; i++) lo = std::min(lo, k[i]);
		if ((lo % width) % 8 != 0) return;
		for (int i = 0; i < 4096; i++) {
			const int32_t d = k[i] - lo;
			const int dy = d / width, dx = d % width;
			if (dx >= 64 || dy >= 64) return;  // not an aligned 64x64 square
			cur[i] = (uint16_t)(dy * 128 + (((dx >> 3) ^ (dy & 7)) << 4) + (dx & 7) * 2);
		}
		int which = -1;
		for (size_t p = 0; p < pats.size(); p++)
			if (pats[p] == cur) { which = (int)p; break; }
		if (which < 0) {
			if ((int)pats.size() == TILE_MAX_ORIENT) return;
			pats.push_back(cur);
			which = (int)pats.size() - 1;
		}
		org[ti] = (uint32_t)lo;
		orient[ti] = (uint8_t)which;
	}
	std::vector<uint16_t> flat;
	for (auto &p : pats) flat.insert(flat.end(), p.begin(), p.end());
	if (hipMalloc(&t.d_org, nt * sizeof(uint32_t)) != hipSuccess) return;
	if (hipMalloc(&t.d_orient, nt) != hipSuccess) return;
	if (hipMalloc(&t.d_pat, flat.size() * sizeof(uint16_t)) != hipSuccess) return;
	if (hipMemcpy(t.d_org, org.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMemcpy(t.d_orient, orient.data(), nt, hipMemcpyHostToDevice) != hipSuccess) return;
	if (hipMemcpy(t.d_pat, flat.data(), flat.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return;
	t.n_tiles = nt;
	t.n_orient = (int)pats.size();
	t.tiled = true;
	build_pipe_tables(O, width, org, orient, t);
}

int get_tables(int width, int height, const ShapeTables **out)
{
	const auto key = std::make_pair(width, height);
	{
		std::lock_guard<std::mutex> lkl(g_mu_lut);
		auto it = g_ctx.luts.find(key);
		if (it != g_ctx.luts.end()) { *out = &it->second; return CCT_OK; }
	}
	// a new shape: the traversal is generated on the host with no lock held; the device allocations and the synchronous
	// copies of the tables run with nothing else of the library in flight (host.h), and g_mu_lut is taken INSIDE the
	// exclusive section (the other order would invert against a thread that holds g_mu_lut and asks for the section)
	const size_t N = (size_t)width * height;
	std::vector<int32_t> host(N ? N : 1);
	if (!gilbert_table(width, height, host.data())) return fail(CCT_E_SHAPE, "traversal generation failed for %dx%d", width, height);
	host.resize(N);
	return exclusive_section([&]() -> int {
		std::lock_guard<std::mutex> lkl(g_mu_lut);
		auto it = g_ctx.luts.find(key);
		if (it != g_ctx.luts.end()) { *out = &it->second; return CCT_OK; }  // another thread built it meanwhile
		ShapeTables t;
		HIP_TRY(hipMalloc(&t.d_lut, (N ? N : 1) * sizeof(int32_t)));
		HIP_TRY(hipMemcpy(t.d_lut, host.data(), N * sizeof(int32_t), hipMemcpyHostToDevice));
		build_tile_tables(host, width, t);
		auto ins = g_ctx.luts.emplace(key, t);
		*out = &ins.first->second;
		return CCT_OK;
	});
}

int get_lut(int width, int height, const int32_t **out)
{
	const ShapeTables *t;
	int rc = get_tables(width, height, &t);
	if (rc) return rc;
	*out = t->d_lut;
	return CCT_OK;
}

bool bs_ok(int bs) { return bs == 4 || bs == 8 || bs == 16 || bs == 32 || bs == 64; }

int check_shape(int n, int width, int height, int bs)
{
	if (n < 0) return fail(CCT_E_ARG, "negative batch size");
	if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
		return fail(CCT_E_ARG, "shape %dx%d outside the 16-bit header fields", width, height);
	if (!bs_ok(bs)) return fail(CCT_E_ARG, "block_size %d not supported by the HIP path (4, 8, 16, 32, 64)", bs);
	const int64_t N = (int64_t)width * height;
	if (N % bs != 0) return fail(CCT_E_SHAPE, "cannot reshape array of size %lld into blocks of %d", (long long)N, bs);
	if (N >= (int64_t)1 << 30) return fail(CCT_E_ARG, "slices of 2^30 pixels or more are not supported");
	return CCT_OK;
}

// Persistent host thread team: fn(i) for i in [0, n).  Spawning 256 threads per call costs more than the
// INFLATE work they do, so workers are kept (and re-created in a forked child, where they do not exist).
class HostTeam {
public:
	template <class F>
	void run(int n, int threads, F fn)
	{
		if (n <= 0) return;
		const int nt = std::max(1, std::min(threads, n));
		if (nt == 1) { for (int i = 0; i < n; i++) fn(i); return; }
		std::unique_lock<std::mutex> lk(m_);
		if (pid_ != getpid()) {  // forked child: the parent's workers are not here
			nworkers_ = 0;
			pid_ = getpid();
		}
		while (nworkers_ < nt - 1) { std::thread([this] { loop(); }).detach(); nworkers_++; }  // daemon workers
		job_ = [&fn](int i) { fn(i); };
		n_ = n; next_.store(0); pending_ = std::min(nt - 1, nworkers_); want_ = pending_; gen_++;
		cv_.notify_all();
		lk.unlock();
		for (int i; (i = next_.fetch_add(1)) < n;) fn(i);  // the caller works too
		lk.lock();
		done_.wait(lk, [this] { return pending_ == 0; });
		job_ = nullptr;
	}
private:
	void loop()
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m_);
		for (;;) {
			cv_.wait(lk, [&] { return gen_ != seen && want_ > 0; });
			seen = gen_;
			want_--;
			auto job = job_;
			const int n = n_;
			lk.unlock();
			for (int i; (i = next_.fetch_add(1)) < n;) job(i);
			lk.lock();
			if (--pending_ == 0) done_.notify_all();
		}
	}
	std::mutex m_;
	std::condition_variable cv_, done_;
	int nworkers_ = 0;
	std::function<void(int)> job_;
	std::atomic<int> next_{0};
	int n_ = 0, pending_ = 0, want_ = 0;
	uint64_t gen_ = 0;
	pid_t pid_ = getpid();
};
// never destroyed: detached workers may still wait on the condition variables at process exit
HostTeam &g_team_enc = *new HostTeam, &g_team_dec = *new HostTeam;  // encode- and decode-side host phases may overlap

template <class F>
void parallel_for(int n, int threads, F fn, HostTeam &team = g_team_enc)
{
	team.run(n, threads, fn);
}

int encode_payload_locked(EncSlot &E, hipStream_t st, const uint16_t *d_images, int n, int width, int height, int bs, uint32_t flags,
                          int eof, uint8_t *d_payload, size_t stride, uint32_t *d_sizes, uint32_t *d_status,
                          cct_slice_stats *d_stats, uint8_t *d_roles)
{
	const int N = width * height, NB = N / bs;
	if (stride < cct_payload_stride(width, height, bs)) return fail(CCT_E_CAP, "payload stride %zu too small", stride);
	if (n == 0) return CCT_OK;
	EncArgs a{};
	a.images = d_images;
	a.lut = nullptr;
	if (flags & CCT_FLAG_FRACTAL) { int rc = get_lut(width, height, &a.lut); if (rc) return rc; }
	a.N = N; a.NB = NB; a.eof = eof; a.flags = flags;
	a.payload = d_payload; a.stride = stride; a.sizes = d_sizes; a.status = d_status;
	a.stats = reinterpret_cast<uint32_t *>(d_stats); a.roles_out = d_roles;
	a.dbg_skip = (uint32_t)g_ctx.dbg_skip;
	bool role_in_lds = true;
	(void)enc_lds_bytes(NB, &role_in_lds);
	const size_t per = (size_t)n * NB;
	int rc;
	if (!role_in_lds) { if ((rc = E.e_role.ensure(per))) return rc; a.ws_role = (uint8_t *)E.e_role.p; }
	if ((rc = E.e_lidx.ensure(per * 4))) return rc;
	if ((rc = E.e_lmask.ensure(per * 8))) return rc;
	if ((rc = E.e_lcur.ensure(per))) return rc;
	a.ws_lidx = (uint32_t *)E.e_lidx.p; a.ws_lmask = (uint64_t *)E.e_lmask.p; a.ws_lcur = (uint8_t *)E.e_lcur.p;
	const ShapeTables *tb = nullptr;
	if ((flags & CCT_FLAG_FRACTAL) && bs == 16 && g_ctx.use_tiles) { if ((rc = get_tables(width, height, &tb))) return rc; }
	// a tile-path launch sequence (memset nodes + kernels), replayed as a graph from the second call with the same arguments
	// on: no dispatch gaps between its nodes, fewer host calls
	auto launch_or_replay = [&](const void *args, size_t args_bytes, int tag, const std::function<hipError_t()> &launch) -> int {
		if (!g_ctx.use_graph) { HIP_TRY(launch()); return CCT_OK; }
		std::vector<uint8_t> key(args_bytes + 2 * sizeof(int));
		memcpy(key.data(), args, args_bytes);
		memcpy(key.data() + args_bytes, &n, sizeof(int));
		memcpy(key.data() + args_bytes + sizeof(int), &tag, sizeof(int));
		EncSlot::PipeGraph *pg = nullptr;
		for (auto &gr : E.p_graphs) if (gr.key == key) pg = &gr;
		if (!pg) {
			if (E.p_graphs.size() >= 6) {  // forget the least recently used argument set
				size_t old = 0;
				for (size_t i = 1; i < E.p_graphs.size(); i++) if (E.p_graphs[i].last_use < E.p_graphs[old].last_use) old = i;
				exclusive_section([&]() -> int {
					if (E.p_graphs[old].exec) (void)hipGraphExecDestroy(E.p_graphs[old].exec);
					if (E.p_graphs[old].graph) (void)hipGraphDestroy(E.p_graphs[old].graph);
					return 0;
				});
				E.p_graphs.erase(E.p_graphs.begin() + (long)old);
			}
			E.p_graphs.emplace_back();
			E.p_graphs.back().key = key;
			E.p_graphs.back().last_use = ++E.p_clock;
			HIP_TRY(launch());  // first sight: plain launches (also sets the kernel attributes)
			return CCT_OK;
		}
		pg->last_use = ++E.p_clock;
		if (!pg->exec) {
			const int crc = exclusive_section([&]() -> int {  // nothing else of the library runs during a capture (host.h)
				HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
				hipError_t le = launch();
				hipError_t ce = hipStreamEndCapture(st, &pg->graph);
				if (le != hipSuccess) return fail(CCT_E_DEVICE, "tile-path capture: %s", hipGetErrorString(le));
				HIP_TRY(ce);
				HIP_TRY(hipGraphInstantiate(&pg->exec, pg->graph, nullptr, nullptr, 0));
				return CCT_OK;
			});
			if (crc) return crc;
		}
		HIP_TRY(hipGraphLaunch(pg->exec, st));
		return CCT_OK;
	};
	// Choice among the tile paths (option "tile_path"): 1 = default: the streaming kernel (encode_stream.hip) wherever it
	// applies; 4 forces it, 3 the four-kernel pipeline of round 2, 2 the one-workgroup-per-slice tile kernel of round 1,
	// 0 the generic table-gather kernel.
	const bool pipe_ok = tb && tb->tiled && tb->pipe && NB <= PIPE_MAX_NB && !g_ctx.dbg_skip;
	const int tpg = g_ctx.stream_tpg;
	const int gps = tb ? (tb->n_tiles + tpg - 1) / tpg : 0;
	if (pipe_ok && (g_ctx.use_tiles == 1 || g_ctx.use_tiles == 4)) {
		if ((rc = E.e_pairrec.ensure((size_t)n * (NB / 2) * PIPE_PAIR_REC))) return rc;
		if ((rc = E.e_hand.ensure(stream_ws_bytes(n, gps)))) return rc;
		StreamArgs sa{};
		sa.e = a;
		sa.tiles = tb->tiles; sa.ptab = tb->d_ptab; sa.htab = tb->d_htab; sa.otab = tb->d_otab; sa.ttab = tb->d_ttab;
		sa.n_tiles = tb->n_tiles; sa.row_pitch = width; sa.gps = gps; sa.tpg = tpg;
		{ static const char *dbg = getenv("CCT_STREAM_DBG"); sa.dbg = dbg ? atoi(dbg) : 0; }
		sa.hand = (uint64_t *)E.e_hand.p; sa.ticket = (uint32_t *)(sa.hand + (size_t)n * gps * 4);
		sa.spill_mask = (uint64_t *)E.e_lmask.p; sa.spill_idx = (uint16_t *)E.e_lidx.p; sa.pairrec = (uint8_t *)E.e_pairrec.p;
		g_ctx.last_path = 3;
		static const bool sstamps = getenv("CCT_STREAM_STAMPS") != nullptr;
		(void)sstamps;
		// two memsets and one kernel: launched plainly.  Replayed as a graph the three nodes took 0.20 ms in the bench against 0.15
		// (profiles/r03_graph_ab.log): a graph pays between its nodes what it saves on the host, and there is nothing to save here
		HIP_TRY(launch_encode_stream(sa, n, st));
		return CCT_OK;
	}
	if (pipe_ok && g_ctx.use_tiles == 3) {
		const int NT = tb->n_tiles;
		if ((rc = E.e_role.ensure(per))) return rc;
		if ((rc = E.e_toff.ensure((size_t)n * (2 * NT + 1) * 4))) return rc;
		if ((rc = E.e_pairrec.ensure((size_t)n * (NB / 2) * PIPE_PAIR_REC))) return rc;
		if ((rc = E.e_spill.ensure(per * 4))) return rc;
		if ((rc = E.e_tflag.ensure((size_t)n * (NT + 1) * 4 + (size_t)n * NT * 4))) return rc;
		PipeArgs pa{};
		pa.e = a;
		pa.tiles = tb->tiles; pa.ptab = tb->d_ptab; pa.ptab2 = tb->d_ptab2; pa.btab = tb->d_btab; pa.otab = tb->d_otab;
		pa.ttab = tb->d_ttab;
		pa.n_orient = tb->n_orient; pa.n_tiles = NT; pa.row_pitch = width;
		pa.ssz = (uint8_t *)E.e_lcur.p; pa.mask = (uint64_t *)E.e_lmask.p; pa.roles = (uint8_t *)E.e_role.p;
		pa.spec = (uint32_t *)E.e_lidx.p; pa.toff = (uint32_t *)E.e_toff.p; pa.pairrec = (uint8_t *)E.e_pairrec.p;
		pa.spill_idx = (uint32_t *)E.e_spill.p;
		pa.tflag = (uint32_t *)E.e_tflag.p; pa.tcount = pa.tflag + (size_t)n * NT; pa.tlist = pa.tcount + n;
		PipeTune tune{g_ctx.pipe_tpw, g_ctx.pipe_timing ? g_ctx.pipe_us : nullptr};
		g_ctx.last_path = 1;
		static const bool stamps = getenv("CCT_PIPE_STAMPS") != nullptr;
		if (!g_ctx.pipe_timing && !stamps) return launch_or_replay(&pa, sizeof pa, tune.tpw, [&]() { return launch_encode_pipe(pa, n, st, &tune); });
		HIP_TRY(launch_encode_pipe(pa, n, st, &tune));
		return CCT_OK;
	}
	if (tb && tb->tiled) {
		bool tile_role_in_lds = true;
		(void)enc_tiles_lds_bytes(NB, &tile_role_in_lds);
		if (!tile_role_in_lds) { if ((rc = E.e_role.ensure(per))) return rc; a.ws_role = (uint8_t *)E.e_role.p; }
		else a.ws_role = nullptr;
		TileEncArgs ta{};
		ta.e = a;
		ta.tile_org = tb->d_org; ta.tile_orient = tb->d_orient; ta.patterns = tb->d_pat;
		ta.n_orient = tb->n_orient; ta.n_tiles = tb->n_tiles; ta.row_pitch = width;
		HIP_TRY(launch_encode_tiles(ta, n, st));
		g_ctx.last_path = 2;
		return CCT_OK;
	}
	HIP_TRY(launch_encode(a, n, bs, g_ctx.wg_threads, st));
	g_ctx.last_path = 0;
	return CCT_OK;
}

// Payload bytes one DEFLATE pass takes (its workspaces need 40 bytes per payload byte: 43 GB at this bound, of 288).  2^28
// in round 1; at 1024x1024 that made passes of 127 slices, and the kernels that give a slice one workgroup (both sort
// passes, the run list, the block walk) left half of the 256 CUs idle.
constexpr size_t DEFLATE_PASS_BYTES = (size_t)1 << 30;

// DEFLATE (zlib level 9 stream) of n device-resident byte strings on the device; output slice i =
// 13 header bytes + zlib stream at d_out + i*out_stride (E.z_out), sizes in E.z_outsizes.
int deflate_locked(EncSlot &E, const uint8_t *d_in, size_t in_stride, const uint32_t *d_in_sizes, int n, const uint8_t header13[13],
                   size_t out_stride)
{
	if (in_stride % 256 != 0 || out_stride % 4 != 0) return fail(CCT_E_ARG, "deflate strides must be multiples of 256 / 4");
	const size_t EB = (size_t)n * in_stride;
	if (EB >= ((size_t)1 << 32)) return fail(CCT_E_ARG, "deflate batch of %zu bytes exceeds the 4 GiB sort limit; split the batch", EB);
	const int max_blocks = (int)(in_stride / 16383 + 2);
	int rc;
	if ((rc = E.z_vals_in.ensure(EB * 8))) return rc;   // records between the two sort passes, then run_ends + run_len
	if ((rc = E.z_vals_out.ensure(EB * 8))) return rc;  // sorted records
	if ((rc = E.z_mr.ensure(EB * 8))) return rc;
	if ((rc = E.z_runs.ensure(EB * 4))) return rc;
	if ((rc = E.z_rec.ensure(EB * 4))) return rc;
	if ((rc = E.z_sym.ensure(EB * 4))) return rc;
	if ((rc = E.z_exitp.ensure(EB * 4))) return rc;
	if ((rc = E.z_exitc.ensure(EB * 4))) return rc;
	if ((rc = E.z_bentry.ensure(EB / 64 * 4))) return rc;
	if ((rc = E.z_bsym.ensure(EB / 64 * 4))) return rc;
	const int run_chunks = (int)(in_stride / 1784 + 1);  // chunks of dfl_run_len_kernel (RUNLEN_OUT positions) per slice
	if ((rc = E.z_small.ensure((size_t)n * (9 + 384 + 2 * (size_t)run_chunks) * 4))) return rc;
	if ((rc = E.z_bend.ensure((size_t)n * max_blocks * 4))) return rc;
	if ((rc = E.z_meta.ensure((size_t)n * max_blocks * sizeof(BlockMeta)))) return rc;
	if ((rc = E.z_tables.ensure((size_t)n * max_blocks * sizeof(BlockTables)))) return rc;
	if ((rc = E.z_out.ensure((size_t)n * out_stride))) return rc;
	if ((rc = E.z_outsizes.ensure((size_t)n * 4))) return rc;
	// (cutting the batch into slice ranges that run concurrently on streams of their own was tried and removed: the
	// cross-stream dependencies cost more than the overlap returned, here as in the transform+pack stage)
	const size_t tmp = (deflate_sort_temp_bytes((size_t)n * in_stride, n) + 511) & ~(size_t)255;
	if ((rc = E.z_sorttmp.ensure(tmp + 256))) return rc;
	const bool use_fork = g_ctx.deflate_fork != 0;
	if (use_fork && !E.stream_side) {
		// created on first use (see GPU_MAX_HW_QUEUES in ensure_ctx: streams are not free)
		const int crc = exclusive_section([&]() -> int { HIP_TRY(hipStreamCreateWithFlags(&E.stream_side, hipStreamNonBlocking)); return CCT_OK; });
		if (crc) return crc;
	}
	DeflateArgs a{};
	a.in = d_in; a.in_stride = in_stride; a.in_sizes = d_in_sizes;
	a.rec_in = (uint64_t *)E.z_vals_in.p; a.rec_out = (uint64_t *)E.z_vals_out.p;
	a.pos_mask = (g_ctx.compact_recs && in_stride < ((size_t)1 << 22)) ? (1u << 22) - 1u : 0xFFFFFFFFu;
	uint32_t *small = (uint32_t *)E.z_small.p;
	a.seg_begin = small; a.seg_end = small + n; a.total_syms = small + 2 * n; a.postloop_lit = small + 3 * n;
	a.n_blocks = small + 4 * n; a.adler = small + 5 * n; a.heavy_count = small + 6 * n; a.deep_count = small + 7 * n; a.run_end_count = small + 8 * n; a.sort_hist = small + 9 * n;
	a.run_counts = small + (9 + 384) * (size_t)n; a.run_chunks = run_chunks;
	// the tag of a pass must not meet a record of 16383 passes ago: clear records and counter well before it comes round (and
	// whenever the buffer is new); a pass that failed on the way may have left the two counts a few apart, hence the margin
	if ((rc = E.z_gen.ensure(256))) return rc;
	if (E.z_mr_cleared != E.z_mr.p || E.z_mr_cleared_cap != E.z_mr.cap || E.z_gen_passes >= 16000u) {
		HIP_TRY(hipMemsetAsync(E.z_mr.p, 0, E.z_mr.cap, E.stream));
		HIP_TRY(hipMemsetAsync(E.z_gen.p, 0, 256, E.stream));
		E.z_mr_cleared = E.z_mr.p; E.z_mr_cleared_cap = E.z_mr.cap; E.z_gen_passes = 0;
	}
	E.z_gen_passes++;
	a.gen = (uint32_t *)E.z_gen.p;
	a.mr = E.z_mr.p; a.heavy_list = (uint32_t *)E.z_rec.p; a.sym = (uint32_t *)E.z_sym.p;
	a.run_ends = (uint32_t *)E.z_runs.p;  // 4 EB of their own: the lists are built while the sort runs (launch_deflate)
	a.run_len = (uint16_t *)E.z_sym.p;  // 2 EB, written before the sort and dead before the symbols are
	a.rec32 = (uint32_t *)E.z_exitp.p; a.exit_pos = (uint32_t *)E.z_exitc.p; a.exit_cnt = (uint32_t *)E.z_rec.p;  // the heavy/deep queues are dead once dfl_rec_kernel runs
	a.blk_entry = (uint32_t *)E.z_bentry.p; a.blk_symbase = (uint32_t *)E.z_bsym.p;
	a.blk_end = (uint32_t *)E.z_bend.p;
	a.block_meta = (BlockMeta *)E.z_meta.p; a.block_tables = (BlockTables *)E.z_tables.p;
	a.max_blocks = max_blocks;
	a.out = (uint8_t *)E.z_out.p; a.out_stride = out_stride; a.out_sizes = (uint32_t *)E.z_outsizes.p;
	memcpy(a.header13, header13, 13);
	if (g_ctx.use_graph) {
		// whole batch on the main stream, as a graph keyed by everything the launches depend on
		std::vector<uint8_t> key(sizeof(DeflateArgs) + 2 * sizeof(int));
		memcpy(key.data(), &a, sizeof(DeflateArgs));
		memcpy(key.data() + sizeof(DeflateArgs), &n, sizeof(int));
		const int fork_key = use_fork ? 1 : 0;
		memcpy(key.data() + sizeof(DeflateArgs) + sizeof(int), &fork_key, sizeof(int));
		EncSlot::ZGraph *zg = nullptr;
		for (auto &g : E.z_graphs) if (g.key == key) zg = &g;
		if (!zg) {
			const int crc = exclusive_section([&]() -> int {  // nothing else of the library runs during a capture (host.h)
				if (E.z_graphs.size() >= 4) {  // forget the least recently used argument set
					size_t old = 0;
					for (size_t i = 1; i < E.z_graphs.size(); i++) if (E.z_graphs[i].last_use < E.z_graphs[old].last_use) old = i;
					if (E.z_graphs[old].exec) (void)hipGraphExecDestroy(E.z_graphs[old].exec);
					if (E.z_graphs[old].graph) (void)hipGraphDestroy(E.z_graphs[old].graph);
					E.z_graphs.erase(E.z_graphs.begin() + (long)old);
				}
				EncSlot::ZGraph g;
				HIP_TRY(hipStreamBeginCapture(E.stream, hipStreamCaptureModeThreadLocal));
				hipError_t le = launch_deflate(a, n, E.z_sorttmp.p, tmp, E.stream, use_fork ? E.stream_side : nullptr, E.ev_fork);
				hipError_t ce = hipStreamEndCapture(E.stream, &g.graph);
				if (le != hipSuccess) { if (g.graph) (void)hipGraphDestroy(g.graph); return fail(CCT_E_DEVICE, "DEFLATE capture: %s", hipGetErrorString(le)); }
				HIP_TRY(ce);
				const hipError_t ie = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
				if (ie != hipSuccess) { (void)hipGraphDestroy(g.graph); return fail(CCT_E_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
				g.key = key;
				E.z_graphs.push_back(std::move(g));
				return CCT_OK;
			});
			if (crc) return crc;
			zg = &E.z_graphs.back();
		}
		zg->last_use = ++E.z_clock;
		HIP_TRY(hipGraphLaunch(zg->exec, E.stream));
		return CCT_OK;
	}
	HIP_TRY(launch_deflate(a, n, E.z_sorttmp.p, tmp, E.stream, use_fork ? E.stream_side : nullptr, E.ev_fork));
	return CCT_OK;
}

int decode_payload_locked(DecSlot &D, const uint8_t *d_payload, size_t stride, const uint32_t *d_sizes, int n, int width,
                          int height, int bs, int fractal, uint16_t *d_images, uint32_t *d_status, hipStream_t st)
{
	const int N = width * height, NB = N / bs;
	if (stride % 16 != 0) return fail(CCT_E_ARG, "payload stride must be a multiple of 16");
	if (n == 0) return CCT_OK;
	DecArgs a{};
	a.payload = d_payload; a.stride = stride; a.sizes = d_sizes;
	a.lut = nullptr;
	a.n_tiles = 0; a.n_orient = 0; a.width = width;
	if (fractal) {
		const ShapeTables *tb;
		int rc0 = get_tables(width, height, &tb);
		if (rc0) return rc0;
		a.lut = tb->d_lut;
		if (tb->tiled && g_ctx.use_tiles) {
			a.tile_org = tb->d_org; a.tile_orient = tb->d_orient; a.patterns = tb->d_pat;
			a.n_tiles = tb->n_tiles; a.n_orient = tb->n_orient;
		}
	}
	a.N = N; a.NB = NB; a.images = d_images; a.status = d_status;
	const size_t per = (size_t)n * NB, jper = (size_t)n * ((size_t)NB / 2 + 1);
	int rc;
	if ((rc = D.d_role.ensure(per))) return rc;
	if ((rc = D.d_slot.ensure(per * 4))) return rc;
	if ((rc = D.d_jord.ensure(jper * 4))) return rc;
	if ((rc = D.d_jval.ensure(jper))) return rc;
	a.ws_role = (uint8_t *)D.d_role.p; a.ws_slot = (uint32_t *)D.d_slot.p;
	a.ws_jord = (uint32_t *)D.d_jord.p; a.ws_jval = (uint8_t *)D.d_jval.p;
	a.pcache_steps = (int)(stride / ((size_t)g_ctx.wg_threads * DEC_SEG) + 2);
	if ((rc = D.d_pcache.ensure((size_t)n * a.pcache_steps * g_ctx.wg_threads * sizeof(uint2)))) return rc;
	a.ws_pcache = (uint2 *)D.d_pcache.p;
	HIP_TRY(launch_decode(a, n, bs, g_ctx.wg_threads, st));
	if (st != D.stream) {
		// the kernel is still queued on the caller's stream when the slot is released: whatever takes the slot next runs on
		// D.stream and must not touch d_role / d_slot / d_jord / d_pcache before this launch is through
		HIP_TRY(hipEventRecord(D.ev_ws, st));
		HIP_TRY(hipStreamWaitEvent(D.stream, D.ev_ws, 0));
	}
	return CCT_OK;
}

}  // namespace
}  // namespace cct

using namespace cct;

extern "C" {

int cct_version(void) { return CCT_ABI_VERSION; }
const char *cct_last_error(void) { return g_err; }

int cct_init(int device)
{
	std::lock_guard<std::mutex> lk(g_mu);
	return ensure_ctx(device);
}

int cct_shutdown(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	std::lock_guard<std::mutex> lk1(g_mu1);
	std::lock_guard<std::mutex> lkd0(g_mu_dec[0]);
	std::lock_guard<std::mutex> lkd1(g_mu_dec[1]);
	if (!g_ctx.ready || g_ctx.pid != getpid()) { reset_ctx(); return CCT_OK; }
	(void)hipSetDevice(g_ctx.device);
	(void)hipStreamSynchronize(g_ctx.stream);
	(void)hipStreamSynchronize(g_ctx.stream_dec);
	for (int k = 0; k < N_ENC_SLOTS; k++) {
		EncSlot &E = g_enc[k];
		if (E.stream) (void)hipStreamSynchronize(E.stream);
		if (E.stream_copy) (void)hipStreamSynchronize(E.stream_copy);
		for (auto &g : E.z_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
		for (auto &g : E.p_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
		for (int i = 0; i < E.n_bufs; i++) E.all_bufs[i]->release();
		hipEvent_t evs[] = {E.ev_k0s[0], E.ev_k1s[0], E.ev_z0s[0], E.ev_z1s[0], E.ev_k0s[1], E.ev_k1s[1], E.ev_z0s[1], E.ev_z1s[1],
		                    E.ev_small[0], E.ev_small[1], E.ev_pack[0], E.ev_pack[1], E.ev_copied[0], E.ev_copied[1],
		                    E.ev_fork[0], E.ev_fork[1], E.ev_fork[2], E.ev_fork[3]};
		for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
		if (E.stream_copy) (void)hipStreamDestroy(E.stream_copy);
		if (E.stream_side) { (void)hipStreamSynchronize(E.stream_side); (void)hipStreamDestroy(E.stream_side); }
		if (k > 0 && E.stream) (void)hipStreamDestroy(E.stream);
	}
	comm_release();
	if (g_gate) { (void)hipFree(g_gate); g_gate = nullptr; }
	for (auto &kv : g_ctx.luts) {
		ShapeTables &t = kv.second;
		void *ptrs[] = {t.d_lut, t.d_org, t.d_orient, t.d_pat, t.d_ptab, t.d_ptab2, t.d_btab, t.d_otab, t.d_ttab, t.d_htab};
		for (void *p : ptrs) if (p) (void)hipFree(p);
	}
	for (int k = 0; k < N_DEC_SLOTS; k++) {
		DecSlot &D = g_dec[k];
		if (D.stream) (void)hipStreamSynchronize(D.stream);
		for (int i = 0; i < D.n_bufs; i++) D.all_bufs[i]->release();
		hipEvent_t evs[] = {D.ev_d0, D.ev_d1, D.ev_k_dec0, D.ev_k_dec1, D.ev_ws};
		for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
		if (k > 0 && D.stream) (void)hipStreamDestroy(D.stream);
	}
	(void)hipStreamDestroy(g_ctx.stream);
	(void)hipStreamDestroy(g_ctx.stream_dec);
	reset_ctx();
	return CCT_OK;
}

int cct_device_info(char *name, size_t name_cap, int *compute_units, uint64_t *hbm_bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, g_ctx.device));
	if (name && name_cap) {
		// the runtime of this image leaves hipDeviceProp_t::name empty for the MI355X: ask hipDeviceGetName, then fall back
		char dn[256] = "";
		if (prop.name[0]) snprintf(dn, sizeof dn, "%s", prop.name);
		else if (hipDeviceGetName(dn, (int)sizeof dn, g_ctx.device) != hipSuccess || !dn[0]) {
			(void)hipGetLastError();
			snprintf(dn, sizeof dn, "gfx950 device, %d CUs", prop.multiProcessorCount);
		}
		snprintf(name, name_cap, "%s (%s)", dn, prop.gcnArchName);
	}
	if (compute_units) *compute_units = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
	return CCT_OK;
}

int cct_dev_alloc(void **d_ptr, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	return exclusive_section([&]() -> int { HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1)); return CCT_OK; });
}
int cct_dev_free(void *d_ptr)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	return exclusive_section([&]() -> int {  // hipFree waits for the whole device: not next to a capture (host.h)
		HIP_TRY(hipStreamSynchronize(g_ctx.stream));
		HIP_TRY(hipFree(d_ptr));
		return CCT_OK;
	});
}
int cct_host_alloc(void **h_ptr, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	return exclusive_section([&]() -> int { HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)); return CCT_OK; });
}
int cct_host_free(void *h_ptr)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	return exclusive_section([&]() -> int {
		HIP_TRY(hipStreamSynchronize(g_ctx.stream));
		HIP_TRY(hipHostFree(h_ptr));
		return CCT_OK;
	});
}
// is [p, p + bytes) page-locked host memory (cct_host_alloc / hipHostMalloc / hipHostRegister)?
static bool is_pinned_host(const void *p, size_t bytes)
{
	hipPointerAttribute_t at{};
	if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
	if (at.type != hipMemoryTypeHost) return false;
	hipPointerAttribute_t at2{};
	if (bytes > 1 && hipPointerGetAttributes(&at2, (const uint8_t *)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
	return bytes <= 1 || at2.type == hipMemoryTypeHost;
}
int cct_h2d(void *d_dst, const void *h_src, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g_ctx.stream));
	HIP_TRY(hipStreamSynchronize(g_ctx.stream));
	return CCT_OK;
}
int cct_d2h(void *h_dst, const void *d_src, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
	HIP_TRY(hipStreamSynchronize(g_ctx.stream));
	return CCT_OK;
}
int cct_dev_memset(void *d_dst, int value, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(d_dst, value, bytes, g_ctx.stream));
	HIP_TRY(hipStreamSynchronize(g_ctx.stream));  // complete on return: decode calls run on their own stream
	return CCT_OK;
}
int cct_sync(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(g_ctx.stream));
	for (int k = 1; k < N_ENC_SLOTS; k++) {  // an encode batch call is complete when it returns; this is for symmetry only
		std::lock_guard<std::mutex> lk1(*g_enc[k].mu);
		HIP_TRY(hipStreamSynchronize(g_enc[k].stream));
	}
	return CCT_OK;
}

int cct_event_create(void **ev)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	hipEvent_t e;
	HIP_TRY(hipEventCreate(&e));
	*ev = (void *)e;
	return CCT_OK;
}
int cct_event_record(void *ev)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipEventRecord((hipEvent_t)ev, g_ctx.stream));
	return CCT_OK;
}
int cct_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipEventSynchronize((hipEvent_t)ev_stop));
	HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
	return CCT_OK;
}
int cct_event_destroy(void *ev)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc) return rc;
	HIP_TRY(hipEventDestroy((hipEvent_t)ev));
	return CCT_OK;
}

int cct_curve_table(int width, int height, int32_t *h_out)
{
	if (width <= 0 || height <= 0 || !h_out) return fail(CCT_E_ARG, "bad curve arguments");
	return gilbert_table(width, height, h_out) ? CCT_OK : fail(CCT_E_SHAPE, "traversal generation failed");
}

size_t cct_payload_stride(int width, int height, int block_size)
{
	if (width <= 0 || height <= 0 || block_size <= 0) return 0;
	const size_t N = (size_t)width * height, NB = N / block_size;
	// every pixel a 2-byte token, one jump byte per pair of blocks, EOF, flush padding
	const size_t need = 2 * N + NB / 2 + 1 + 16;
	return (need + 255) & ~(size_t)255;
}

size_t cct_file_bound(int width, int height, int block_size)
{
	const size_t p = cct_payload_stride(width, height, block_size);
	return ((13 + compressBound((uLong)p) + 63) & ~(size_t)63);
}

int cct_encode_payload_dev(const uint16_t *d_images, int n, int width, int height, int block_size, uint32_t flags,
                           int eof_byte, uint8_t *d_payload, size_t payload_stride, uint32_t *d_payload_sizes,
                           uint32_t *d_status, cct_slice_stats *d_stats, uint8_t *d_roles)
{
	std::lock_guard<std::mutex> lk(g_mu);
	ApiCall in_call;
	int rc = check_shape(n, width, height, block_size);
	if (rc) return rc;
	if ((rc = ensure_ctx())) return rc;
	return encode_payload_locked(g_enc[0], g_ctx.stream, d_images, n, width, height, block_size, flags, eof_byte, d_payload, payload_stride,
	                             d_payload_sizes, d_status, d_stats, d_roles);
}

// ---- pieces of cct_encode_batch / cct_encode_batch_packed -------------------------------------------------------------

// take a free encode slot (see EncSlot): the first caller gets slot 0, one that arrives while it is busy slot 1
static EncSlot &acquire_encode_slot(std::unique_lock<std::mutex> &lk)
{
	std::mutex *const slot_mu[N_ENC_SLOTS] = {&g_mu, &g_mu1};
	const int nslots = std::max(1, std::min(g_ctx.enc_slots, N_ENC_SLOTS));
	for (;;) {
		for (int k = 0; k < nslots; k++) {
			std::unique_lock<std::mutex> t(*slot_mu[k], std::try_to_lock);
			if (t.owns_lock()) { lk = std::move(t); return g_enc[k]; }
		}
		std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
}

// core.py:193-210 (big-endian fields, values masked to a byte / 16 bits)
static void make_header13(uint8_t hdr13[13], const char magic[4], int width, int height, int channels, int bytes_per_channel,
                          uint32_t flags)
{
	hdr13[0] = (uint8_t)magic[0]; hdr13[1] = (uint8_t)magic[1]; hdr13[2] = (uint8_t)magic[2]; hdr13[3] = (uint8_t)magic[3];
	hdr13[4] = (uint8_t)((width >> 8) & 0xFF); hdr13[5] = (uint8_t)(width & 0xFF);
	hdr13[6] = (uint8_t)((height >> 8) & 0xFF); hdr13[7] = (uint8_t)(height & 0xFF);
	hdr13[8] = (uint8_t)(channels & 0xFF); hdr13[9] = (uint8_t)(bytes_per_channel & 0xFF);
	hdr13[10] = (flags & CCT_FLAG_FRACTAL) ? 1 : 0;
	hdr13[11] = (flags & CCT_FLAG_SEGMENTATION) ? 1 : 0;
	hdr13[12] = (flags & CCT_FLAG_DEFLATE) ? 1 : 0;
}

// One DEFLATE pass worth of files (nc of them, sizes osz[], in E.z_out at stride zstride) on their way to the caller.
struct FilesOut {
	EncSlot &E;
	int nc;
	const uint32_t *osz;     // file sizes (host)
	size_t zstride;
	size_t exact;            // sum of the sizes
	std::vector<size_t> offs;  // 16-byte aligned offsets of the files in a strided-pack buffer
};

// archive layout, whole batch, page-locked destination: pack into one of two device buffers, hand the copy to the copy
// stream and give the slot back (lk) before waiting for it -- the next batch's kernels start while these files travel
// first half: pack the files of a pass into one of the two device buffers (offsets computed on the device from the sizes, so
// this can be queued right behind the DEFLATE pass, before the host knows the sizes -- the pack used to start a launch
// latency after the host had read them); `cap` = what the buffer must hold
static int pack_enqueue(EncSlot &E, int nc, size_t zstride, size_t cap, unsigned *slot_out)
{
	int rc;
	const unsigned slot = E.pack_slot++ & 1u;
	{
		std::unique_lock<std::mutex> pl(g_pack_mu);
		g_pack_cv.wait(pl, [&] { return !g_pack_busy[&E - g_enc][slot]; });
	}
	if (!E.stream_copy) {
		const int crc = exclusive_section([&]() -> int {
			HIP_TRY(hipStreamCreateWithFlags(&E.stream_copy, hipStreamNonBlocking));
			for (int k = 0; k < 2; k++) {
				HIP_TRY(hipEventCreateWithFlags(&E.ev_pack[k], hipEventDisableTiming));
				HIP_TRY(hipEventCreateWithFlags(&E.ev_copied[k], hipEventDisableTiming));
				HIP_TRY(hipEventRecord(E.ev_copied[k], E.stream_copy));
			}
			return CCT_OK;
		});
		if (crc) return crc;
	}
	if (E.z_packed2[slot].cap < cap) HIP_TRY(hipEventSynchronize(E.ev_copied[slot]));  // about to be reallocated
	if ((rc = E.z_packed2[slot].ensure(cap))) return rc;
	if ((rc = E.z_packoffs.ensure((size_t)(nc + 1) * 8))) return rc;
	HIP_TRY(hipStreamWaitEvent(E.stream, E.ev_copied[slot], 0));  // the copy out of this buffer two calls ago
	HIP_TRY(launch_pack((const uint8_t *)E.z_out.p, zstride, (const uint32_t *)E.z_outsizes.p, nc,
	                    (uint64_t *)E.z_packoffs.p, (uint8_t *)E.z_packed2[slot].p, 1, E.stream));
	HIP_TRY(hipEventRecord(E.ev_pack[slot], E.stream));
	*slot_out = slot;
	return CCT_OK;
}

// archive layout, whole batch, page-locked destination: pack into one of two device buffers (unless pack_enqueue has done so:
// packed_slot >= 0), hand the copy to the copy stream and give the slot back (the caller does) before waiting for it -- the
// next batch's kernels start while these files travel
static int files_to_pinned_archive_async(FilesOut &f, uint8_t *h_dst, int packed_slot, hipEvent_t *done)
{
	EncSlot &E = f.E;
	int rc;
	unsigned slot = (unsigned)packed_slot;
	if (packed_slot < 0 && (rc = pack_enqueue(E, f.nc, f.zstride, f.offs[f.nc] + 16, &slot))) return rc;  // (the pack kernel works in 16-byte units)
	HIP_TRY(hipStreamWaitEvent(E.stream_copy, E.ev_pack[slot], 0));
	HIP_TRY(hipMemcpyAsync(h_dst, E.z_packed2[slot].p, f.exact, hipMemcpyDeviceToHost, E.stream_copy));
	HIP_TRY(hipEventRecord(E.ev_copied[slot], E.stream_copy));
	*done = E.ev_copied[slot];
	return CCT_OK;
}

// archive layout, any destination: exact pack on the device, one copy (through the pinned stage unless the destination is
// page-locked itself, then a threaded memcpy)
static int files_to_archive(FilesOut &f, uint8_t *h_dst, double *t_copied)
{
	EncSlot &E = f.E;
	int rc;
	const size_t cap = f.offs[f.nc] + 16;
	if ((rc = E.z_packed.ensure(cap))) return rc;
	HIP_TRY(launch_pack((const uint8_t *)E.z_out.p, f.zstride, (const uint32_t *)E.z_outsizes.p, f.nc,
	                    (uint64_t *)E.z_packoffs.p, (uint8_t *)E.z_packed.p, 1, E.stream));
	if (is_pinned_host(h_dst, f.exact)) {
		HIP_TRY(hipMemcpyAsync(h_dst, E.z_packed.p, f.exact, hipMemcpyDeviceToHost, E.stream));
		HIP_TRY(hipStreamSynchronize(E.stream));
		*t_copied = now_ms();
		return CCT_OK;
	}
	if ((rc = E.h_stage.ensure(cap))) return rc;
	HIP_TRY(hipMemcpyAsync(E.h_stage.p, E.z_packed.p, f.exact, hipMemcpyDeviceToHost, E.stream));
	HIP_TRY(hipStreamSynchronize(E.stream));
	*t_copied = now_ms();
	const int nt = std::min(g_ctx.zlib_threads, 16);
	const size_t per = (f.exact + nt - 1) / nt;
	const uint8_t *stg = (const uint8_t *)E.h_stage.p;
	const size_t exact = f.exact;
	parallel_for(nt, nt, [&](int t) {
		const size_t lo = (size_t)t * per, hi = std::min(exact, lo + per);
		if (lo < hi) memcpy(h_dst + lo, stg + lo, hi - lo);
	});
	return CCT_OK;
}

// strided layout (file i at h_out + i * out_stride): 16-byte aligned pack on the device, one copy into the pinned stage,
// threaded scatter
static int files_to_strided(FilesOut &f, uint8_t *h_first, size_t out_stride)
{
	EncSlot &E = f.E;
	int rc;
	const size_t packed_cap = f.offs[f.nc];
	if ((rc = E.z_packed.ensure(packed_cap + 16))) return rc;
	if ((rc = E.h_stage.ensure(packed_cap + 16))) return rc;
	HIP_TRY(launch_pack((const uint8_t *)E.z_out.p, f.zstride, (const uint32_t *)E.z_outsizes.p, f.nc,
	                    (uint64_t *)E.z_packoffs.p, (uint8_t *)E.z_packed.p, 0, E.stream));
	HIP_TRY(hipMemcpyAsync(E.h_stage.p, E.z_packed.p, packed_cap, hipMemcpyDeviceToHost, E.stream));
	HIP_TRY(hipStreamSynchronize(E.stream));
	const uint8_t *stg = (const uint8_t *)E.h_stage.p;
	const uint32_t *osz = f.osz;
	const std::vector<size_t> &offs = f.offs;
	parallel_for(f.nc, std::min(g_ctx.zlib_threads, 32), [&](int i) { memcpy(h_first + (size_t)i * out_stride, stg + offs[i], osz[i]); });
	return CCT_OK;
}

// DEFLATE (or none) on the host thread team: payloads come back, libz level 9 per slice (core.py:340), header in front
static int files_on_host(EncSlot &E, int n, size_t stride, const std::vector<uint32_t> &psz, const uint8_t hdr13[13], bool defl,
                         uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes, const uint32_t *h_status)
{
	const double t_copy0 = now_ms();
	uint8_t *stage = (uint8_t *)E.h_stage.p;
	for (int i = 0; i < n; i++) {  // bring back only the bytes each slice produced
		if (h_status[i] & CCT_ST_CAP) return fail(CCT_E_CAP, "slice %d overflowed its payload stride", i);
		HIP_TRY(hipMemcpyAsync(stage + (size_t)i * stride, (uint8_t *)E.e_payload.p + (size_t)i * stride, psz[i],
		                       hipMemcpyDeviceToHost, E.stream));
	}
	HIP_TRY(hipStreamSynchronize(E.stream));
	const double t_defl0 = now_ms();
	tl_d2h_ms = (float)(t_defl0 - t_copy0);
	std::atomic<int> zerr(0);
	parallel_for(n, defl ? g_ctx.zlib_threads : 1, [&](int i) {
		uint8_t *o = h_out + (size_t)i * out_stride;
		memcpy(o, hdr13, 13);
		const uint8_t *pl = stage + (size_t)i * stride;
		if (defl) {  // zlib.compress(data, level=9), core.py:340
			uLongf dl = (uLongf)(out_stride - 13);
			const int zr = compress2(o + 13, &dl, pl, psz[i], 9);
			if (zr != Z_OK) { zerr.store(zr); h_out_sizes[i] = 0; return; }
			h_out_sizes[i] = 13 + (uint32_t)dl;
		} else {
			memcpy(o + 13, pl, psz[i]);
			h_out_sizes[i] = 13 + psz[i];
		}
	});
	tl_deflate_ms = (float)(now_ms() - t_defl0);
	if (zerr.load()) return fail(CCT_E_ZLIB, "compress2 failed (%d)", zerr.load());
	return CCT_OK;
}

static int encode_batch_impl(const uint16_t *images, int images_on_device, int n, int width, int height, int block_size,
                             uint32_t flags, int eof_byte, const char magic[4], int channels, int bytes_per_channel,
                             uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status,
                             uint32_t *h_payload_sizes, cct_slice_stats *h_stats, uint64_t *h_packed_offsets)
{
	// h_packed_offsets != NULL: archive layout -- files back to back in h_out (capacity out_stride bytes in
	// total), h_packed_offsets[n+1]; needs the device DEFLATE path (or deflate off)
	const bool packed = h_packed_offsets != nullptr;
	const double t_call0 = now_ms();
	EncodeInFlight in_flight;  // decode calls that start meanwhile pick the INFLATE geometry that shares the device best
	int rc = check_shape(n, width, height, block_size);
	if (rc) return rc;
	if (!(g_ctx.ready && g_ctx.pid == getpid())) {  // first use in this process: bind the device
		std::lock_guard<std::mutex> lk0(g_mu);
		if ((rc = ensure_ctx())) return rc;
	}
	std::unique_lock<std::mutex> lk;
	EncSlot &E = acquire_encode_slot(lk);
	ApiCall in_call;  // after the slot: a thread waiting for a slot must not keep exclusive sections of the others waiting
	const double t_lock0 = now_ms();
	if ((rc = ensure_ctx())) return rc;
	if (n == 0) return CCT_OK;
	const size_t N = (size_t)width * height;
	const size_t stride = cct_payload_stride(width, height, block_size);
	const bool defl = (flags & CCT_FLAG_DEFLATE) != 0;
	if (!packed && out_stride < (defl ? cct_file_bound(width, height, block_size) : 13 + stride))
		return fail(CCT_E_CAP, "out_stride %zu too small", out_stride);
	if (packed && !(defl && g_ctx.device_deflate))
		return fail(CCT_E_ARG, "packed output needs deflate_compression with the device DEFLATE path");

	const uint16_t *d_img = images;
	if (!images_on_device) {
		if ((rc = E.e_images.ensure((size_t)n * N * 2))) return rc;
		HIP_TRY(hipMemcpyAsync(E.e_images.p, images, (size_t)n * N * 2, hipMemcpyHostToDevice, E.stream));
		d_img = (const uint16_t *)E.e_images.p;
	}
	if ((rc = E.e_payload.ensure((size_t)n * stride))) return rc;
	if ((rc = E.e_sizes.ensure((size_t)n * 4))) return rc;
	if ((rc = E.e_status.ensure((size_t)n * 4))) return rc;
	if ((rc = E.h_stage.ensure((size_t)n * stride))) return rc;
	if ((rc = E.e_stats.ensure((size_t)n * sizeof(cct_slice_stats)))) return rc;
	// (running this stage on a stream of the highest priority while the other slot is in its DEFLATE pass was tried: the
	// kernels took as long as without, waves already resident are not displaced)
	const unsigned par = E.call_parity++ & 1u;
	// queue ahead (see below): possible when the whole call can be queued without the host knowing a size
	const bool qa = packed && defl && g_ctx.device_deflate && g_ctx.queue_ahead && (size_t)n * stride <= DEFLATE_PASS_BYTES &&
	                is_pinned_host(h_out, 1);
	uint32_t *l_psz = nullptr, *l_status = nullptr, *l_osz = nullptr;  // landing places in pinned memory
	cct_slice_stats *l_stats = nullptr;
	if (qa) {
		if ((rc = E.h_small[par].ensure((size_t)n * (12 + sizeof(cct_slice_stats))))) return rc;
		l_psz = (uint32_t *)E.h_small[par].p; l_status = l_psz + n; l_osz = l_status + n;
		l_stats = (cct_slice_stats *)(l_osz + n);
	}
	const hipEvent_t ev_k0 = E.ev_k0s[par], ev_k1 = E.ev_k1s[par], ev_z0 = E.ev_z0s[par], ev_z1 = E.ev_z1s[par];
	float t_dev_deflate_ms = 0;
	HIP_TRY(hipEventRecord(ev_k0, E.stream));
	rc = encode_payload_locked(E, E.stream, d_img, n, width, height, block_size, flags, eof_byte, (uint8_t *)E.e_payload.p, stride,
	                           (uint32_t *)E.e_sizes.p, (uint32_t *)E.e_status.p,
	                           h_stats ? (cct_slice_stats *)E.e_stats.p : nullptr, nullptr);
	if (rc) return rc;
	HIP_TRY(hipEventRecord(ev_k1, E.stream));
	HIP_TRY(launch_gate_bump(g_gate, E.stream));  // decode kernels waiting for this stage to be over may go (cct_decode_batch)
	if (h_stats)
		HIP_TRY(hipMemcpyAsync(qa ? l_stats : h_stats, E.e_stats.p, (size_t)n * sizeof(cct_slice_stats), hipMemcpyDeviceToHost, E.stream));
	std::vector<uint32_t> psz(n);
	DrainOnExit drain(E.stream);  // until the first synchronisation below
	HIP_TRY(hipMemcpyAsync(qa ? l_psz : psz.data(), E.e_sizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, E.stream));
	HIP_TRY(hipMemcpyAsync(qa ? l_status : h_status, E.e_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, E.stream));
	// With DEFLATE on the device and the whole batch in one pass the host does not need sizes or status before
	// the DEFLATE kernels are queued (they read the sizes on the device; a payload cannot outgrow its stride):
	// one host synchronisation less per batch.
	const bool one_pass = defl && g_ctx.device_deflate && (size_t)n * stride <= DEFLATE_PASS_BYTES;
	if (!one_pass) {
		HIP_TRY(hipStreamSynchronize(E.stream));
		drain.disarm();
		HIP_TRY(hipEventElapsedTime(&tl_enc_kernel_ms, ev_k0, ev_k1));
	}
	uint8_t hdr13[13];
	make_header13(hdr13, magic, width, height, channels, bytes_per_channel, flags);
	if (defl && g_ctx.device_deflate) {
		// DEFLATE on the device: zlib.compress(data, level=9) (core.py:340) restated in deflate_kernels.hip
		if (!one_pass)
			for (int i = 0; i < n; i++)
				if (h_status[i] & CCT_ST_CAP) return fail(CCT_E_CAP, "slice %d overflowed its payload stride", i);
		const size_t zstride = cct_file_bound(width, height, block_size);
		// bounded workspaces: at most DEFLATE_PASS_BYTES of payload per pass, in passes of equal size (several DEFLATE
		// kernels give a slice one workgroup: a short last pass would leave most of the chip idle)
		const int chunk_max = (int)std::max<size_t>(1, DEFLATE_PASS_BYTES / stride);
		const int n_passes = (n + chunk_max - 1) / chunk_max;
		const int chunk = (n + n_passes - 1) / n_passes;
		tl_deflate_ms = 0; tl_d2h_ms = 0;
		struct CopyGuard {  // an error return must not leave a copy into the caller's archive in flight
			EncSlot &E; bool armed = false;
			~CopyGuard() { if (armed && E.stream_copy) (void)hipStreamSynchronize(E.stream_copy); }
		} copies_in_flight{E};
		for (int c0 = 0; c0 < n; c0 += chunk) {
			const int nc = std::min(chunk, n - c0);
			HIP_TRY(hipEventRecord(ev_z0, E.stream));
			rc = deflate_locked(E, (const uint8_t *)E.e_payload.p + (size_t)c0 * stride, stride,
			                    (const uint32_t *)E.e_sizes.p + c0, nc, hdr13, zstride);
			if (rc) return rc;
			HIP_TRY(hipEventRecord(ev_z1, E.stream));
			g_gate_passes_issued.fetch_add(1, std::memory_order_relaxed);
			HIP_TRY(launch_gate_bump(g_gate + 1, E.stream));
			int packed_slot = -1;
			if (qa) {  // the common pipelined case: pack queued behind the pass
				unsigned ps = 0;
				if ((rc = pack_enqueue(E, nc, zstride, (size_t)nc * zstride + 16, &ps))) return rc;
				packed_slot = (int)ps;
			}
			uint32_t *osz = h_out_sizes + c0;
			HIP_TRY(hipMemcpyAsync(qa ? l_osz : osz, E.z_outsizes.p, (size_t)nc * 4, hipMemcpyDeviceToHost, E.stream));
			// Queue ahead: everything this call needs from the slot is in the stream now, so the slot goes to the next call
			// BEFORE the host waits for the sizes: that call's kernels are queued behind this one's while they run, and the
			// device no longer idles for a launch latency between one batch's pack and the next batch's first kernel and
			// again before its DEFLATE graph (0.2 ms of a 4.2 ms step, profiles/r03_bench_timeline.log).  From here on the
			// call touches its own events, its packed buffer (g_pack_busy) and the copy stream only.
			struct PackOwner {
				int e = -1; unsigned slot = 0;
				void release() { if (e >= 0) { { std::lock_guard<std::mutex> pl(g_pack_mu); g_pack_busy[e][slot] = false; } g_pack_cv.notify_all(); e = -1; } }
				~PackOwner() { release(); }
			} pack_owner;
			const bool queue_ahead = qa;
			if (queue_ahead) {
				HIP_TRY(hipEventRecord(E.ev_small[par], E.stream));
				{ std::lock_guard<std::mutex> pl(g_pack_mu); g_pack_busy[&E - g_enc][packed_slot] = true; }
				pack_owner.e = (int)(&E - g_enc); pack_owner.slot = (unsigned)packed_slot;
				lk.unlock();
				HIP_TRY(hipEventSynchronize(E.ev_small[par]));
				memcpy(psz.data(), l_psz, (size_t)n * 4); memcpy(h_status, l_status, (size_t)n * 4); memcpy(osz, l_osz, (size_t)nc * 4);
				if (h_stats) memcpy(h_stats, l_stats, (size_t)n * sizeof(cct_slice_stats));
			} else HIP_TRY(hipStreamSynchronize(E.stream));
			drain.disarm();  // nothing queued targets this frame any more (later chunks copy into caller memory and synchronise at once)
			HIP_TRY(hipEventElapsedTime(&t_dev_deflate_ms, ev_z0, ev_z1));
			if (one_pass) {
				HIP_TRY(hipEventElapsedTime(&tl_enc_kernel_ms, ev_k0, ev_k1));
				for (int i = 0; i < n; i++)
					if (h_status[i] & CCT_ST_CAP) return fail(CCT_E_CAP, "slice %d overflowed its payload stride", i);
			}
			tl_deflate_ms += t_dev_deflate_ms;
			const double t_c0 = now_ms();
			FilesOut f{E, nc, osz, zstride, 0, std::vector<size_t>(nc + 1, 0)};
			for (int i = 0; i < nc; i++) { f.exact += osz[i]; f.offs[i + 1] = f.offs[i] + (((size_t)osz[i] + 15) & ~(size_t)15); }
			if (packed_slot < 0 && (rc = E.z_packoffs.ensure((size_t)(nc + 1) * 8))) return rc;
			if (!packed) {
				if ((rc = files_to_strided(f, h_out + (size_t)c0 * out_stride, out_stride))) return rc;
				tl_d2h_ms += (float)(now_ms() - t_c0);
				continue;
			}
			if (c0 == 0) h_packed_offsets[0] = 0;
			const uint64_t at = h_packed_offsets[c0];
			if (at + f.exact > out_stride) return fail(CCT_E_CAP, "packed output needs %zu bytes", (size_t)(at + f.exact));
			for (int i = 0; i < nc; i++) h_packed_offsets[c0 + i + 1] = h_packed_offsets[c0 + i] + osz[i];
			if (is_pinned_host(h_out + at, f.exact)) {
				// page-locked archive: the files of this pass leave on the copy stream while the next pass (or, after the last
				// one, the next batch) already runs its kernels
				hipEvent_t done = nullptr;
				if ((rc = files_to_pinned_archive_async(f, h_out + at, packed_slot, &done))) return rc;
				pack_owner.release();  // the copy out of the buffer is queued: the call after next may pack into it
				copies_in_flight.armed = true;
				tl_d2h_ms += (float)(now_ms() - t_c0);
				if (c0 + nc < n) continue;
				if (h_payload_sizes) memcpy(h_payload_sizes, psz.data(), (size_t)n * 4);
				const float t_defl = t_dev_deflate_ms;
				const double t_unlock = now_ms();
				if (lk.owns_lock()) lk.unlock();  // the slot is free for the next batch; only the copy stream still works for this one
				HIP_TRY(hipEventSynchronize(done));
				copies_in_flight.armed = false;
				if (getenv("CCT_TRACE"))
					fprintf(stderr, "[cct] encode n=%d: waited for a slot %.2f ms, held it %.2f ms (kernel %.2f, deflate %.2f), tail %.2f ms, t=%.2f\n", nc,
					        t_lock0 - t_call0, t_unlock - t_lock0, tl_enc_kernel_ms, t_defl, now_ms() - t_unlock, t_lock0);
				tl_d2h_ms += (float)(now_ms() - t_c0);  // (no lock: a float for cct_last_timings)
				return CCT_OK;
			}
			if (queue_ahead) return fail(CCT_E_ARG, "the archive buffer is page-locked at its start but not over the %zu bytes of this batch", f.exact);
			double t_c1 = t_c0;
			if ((rc = files_to_archive(f, h_out + at, &t_c1))) return rc;
			tl_d2h_ms += (float)(now_ms() - t_c0);
			if (getenv("CCT_TRACE"))
				fprintf(stderr, "[cct] encode n=%d: deflate %.2f ms, pack+d2h %.2f ms, host scatter %.2f ms (%zu bytes)\n", nc,
				        t_dev_deflate_ms, t_c1 - t_c0, now_ms() - t_c1, f.exact);
		}
		if (h_payload_sizes) memcpy(h_payload_sizes, psz.data(), (size_t)n * 4);
		return CCT_OK;
	}
	rc = files_on_host(E, n, stride, psz, hdr13, defl, h_out, out_stride, h_out_sizes, h_status);
	if (h_payload_sizes) memcpy(h_payload_sizes, psz.data(), (size_t)n * 4);
	return rc;
}

int cct_encode_batch(const uint16_t *images, int images_on_device, int n, int width, int height, int block_size,
                     uint32_t flags, int eof_byte, const char magic[4], int channels, int bytes_per_channel,
                     uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status,
                     uint32_t *h_payload_sizes, cct_slice_stats *h_stats)
{
	return encode_batch_impl(images, images_on_device, n, width, height, block_size, flags, eof_byte, magic, channels,
	                         bytes_per_channel, h_out, out_stride, h_out_sizes, h_status, h_payload_sizes, h_stats, nullptr);
}

int cct_encode_batch_packed(const uint16_t *images, int images_on_device, int n, int width, int height, int block_size,
                            uint32_t flags, int eof_byte, const char magic[4], int channels, int bytes_per_channel,
                            uint8_t *h_archive, size_t archive_cap, uint64_t *h_offsets, uint32_t *h_out_sizes,
                            uint32_t *h_status, uint32_t *h_payload_sizes, cct_slice_stats *h_stats)
{
	if (!h_offsets) return fail(CCT_E_ARG, "h_offsets is required");
	return encode_batch_impl(images, images_on_device, n, width, height, block_size, flags, eof_byte, magic, channels,
	                         bytes_per_channel, h_archive, archive_cap, h_out_sizes, h_status, h_payload_sizes, h_stats, h_offsets);
}

int cct_zlib_compress_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, uint8_t *h_out, size_t out_stride,
                            uint32_t *h_out_sizes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	ApiCall in_call;
	if (n < 0) return fail(CCT_E_ARG, "negative batch size");
	int rc = ensure_ctx();
	if (rc) return rc;
	if (n == 0) return CCT_OK;
	EncSlot &E = g_enc[0];
	size_t longest = 0;
	for (int i = 0; i < n; i++) longest = std::max(longest, (size_t)(h_offsets[i + 1] - h_offsets[i]));
	const size_t in_stride = (longest + 16 + 255) & ~(size_t)255;
	const size_t zstride = (13 + compressBound((uLong)in_stride) + 63) & ~(size_t)63;
	if (out_stride < zstride - 13) return fail(CCT_E_CAP, "out_stride %zu too small (need %zu)", out_stride, zstride - 13);
	if ((rc = E.z_in.ensure((size_t)n * in_stride))) return rc;
	if ((rc = E.z_insizes.ensure((size_t)n * 4))) return rc;
	std::vector<uint32_t> isz(n), osz(n);
	DrainOnExit drain(E.stream);
	HIP_TRY(hipMemsetAsync(E.z_in.p, 0, (size_t)n * in_stride, E.stream));
	for (int i = 0; i < n; i++) {
		isz[i] = (uint32_t)(h_offsets[i + 1] - h_offsets[i]);
		if (isz[i])
			HIP_TRY(hipMemcpyAsync((uint8_t *)E.z_in.p + (size_t)i * in_stride, h_in + h_offsets[i], isz[i],
			                       hipMemcpyHostToDevice, E.stream));
	}
	HIP_TRY(hipMemcpyAsync(E.z_insizes.p, isz.data(), (size_t)n * 4, hipMemcpyHostToDevice, E.stream));
	uint8_t hdr13[13] = {0};
	rc = deflate_locked(E, (const uint8_t *)E.z_in.p, in_stride, (const uint32_t *)E.z_insizes.p, n, hdr13, zstride);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(osz.data(), E.z_outsizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, E.stream));
	HIP_TRY(hipStreamSynchronize(E.stream));
	for (int i = 0; i < n; i++) {
		h_out_sizes[i] = osz[i] - 13;
		HIP_TRY(hipMemcpyAsync(h_out + (size_t)i * out_stride, (uint8_t *)E.z_out.p + (size_t)i * zstride + 13, osz[i] - 13,
		                       hipMemcpyDeviceToHost, E.stream));
	}
	HIP_TRY(hipStreamSynchronize(E.stream));
	return CCT_OK;
}

// take a free decode slot (see DecSlot)
static DecSlot &acquire_decode_slot(std::unique_lock<std::mutex> &lk)
{
	const int nslots = std::max(1, std::min(g_ctx.dec_slots, N_DEC_SLOTS));
	for (;;) {
		for (int k = 0; k < nslots; k++) {
			std::unique_lock<std::mutex> t(g_mu_dec[k], std::try_to_lock);
			if (t.owns_lock()) { lk = std::move(t); return g_dec[k]; }
		}
		std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
}

int cct_zlib_decompress_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, uint8_t *h_out, size_t out_stride,
                              uint32_t *h_out_sizes, uint32_t *h_status)
{
	if (n < 0) return fail(CCT_E_ARG, "negative batch size");
	if (out_stride == 0 || (out_stride & 15)) return fail(CCT_E_ARG, "out_stride must be a positive multiple of 16");
	// first use in this process: bind the device BEFORE the slot and the shared lock are taken.  A thread must never wait
	// for g_mu while it counts as a call in flight: a first encode holds g_mu through HIP initialisation and then asks
	// for an exclusive section, which waits for every call in flight to leave
	if (!(g_ctx.ready && g_ctx.pid == getpid())) {
		std::lock_guard<std::mutex> lk(g_mu);
		int rc0 = ensure_ctx();
		if (rc0) return rc0;
	}
	std::unique_lock<std::mutex> lkd;
	DecSlot &D = acquire_decode_slot(lkd);
	ApiCall in_call;
	if (n == 0) return CCT_OK;
	int rc;
	HIP_TRY(hipSetDevice(g_ctx.device));
	const uint64_t a0 = h_offsets[0], a1 = h_offsets[n];
	const size_t abytes = (size_t)(a1 - a0), apad = (abytes + 31) & ~(size_t)15;
	if ((rc = D.d_arch.ensure(apad + 16))) return rc;
	if ((rc = D.d_archoffs.ensure((size_t)(n + 1) * 8))) return rc;
	if ((rc = D.d_zstatus.ensure((size_t)n * 4))) return rc;
	if ((rc = D.d_sizes.ensure((size_t)n * 4))) return rc;
	if ((rc = D.d_payload.ensure((size_t)n * out_stride))) return rc;
	std::vector<uint64_t> rel(n + 1);
	std::vector<uint32_t> zst(n), osz(n);
	for (int i = 0; i <= n; i++) rel[i] = h_offsets[i] - a0;
	hipStream_t st = D.stream;
	DrainOnExit drain(st);
	HIP_TRY(hipMemsetAsync((uint8_t *)D.d_arch.p + (apad > 32 ? apad - 32 : 0), 0, apad > 32 ? 48 : apad + 16, st));
	if (abytes) HIP_TRY(hipMemcpyAsync(D.d_arch.p, h_in + a0, abytes, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(D.d_archoffs.p, rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
	InflateArgs ia{};
	ia.in = (const uint8_t *)D.d_arch.p; ia.in_total = apad;
	ia.offsets = (const uint64_t *)D.d_archoffs.p; ia.skip = 0;
	ia.out = (uint8_t *)D.d_payload.p; ia.out_stride = out_stride;
	ia.out_sizes = (uint32_t *)D.d_sizes.p; ia.status = (uint32_t *)D.d_zstatus.p;
	HIP_TRY(launch_inflate(ia, n, st, inflate_lanes_now()));
	HIP_TRY(hipMemcpyAsync(zst.data(), D.d_zstatus.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(osz.data(), D.d_sizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	int first = CCT_OK;
	for (int i = 0; i < n; i++) {
		h_status[i] = (zst[i] & CCT_ST_ZLIB) ? CCT_E_ZLIB : (zst[i] & CCT_ST_STREAM) ? CCT_E_CAP : CCT_OK;
		h_out_sizes[i] = h_status[i] == CCT_OK ? osz[i] : 0;
		if (h_status[i] == CCT_OK && osz[i])
			HIP_TRY(hipMemcpyAsync(h_out + (size_t)i * out_stride, (uint8_t *)D.d_payload.p + (size_t)i * out_stride, osz[i],
			                       hipMemcpyDeviceToHost, st));
		if (h_status[i] != CCT_OK && first == CCT_OK) {
			first = (int)h_status[i];
			fail(first, "stream %d: %s", i, first == CCT_E_ZLIB ? "invalid DEFLATE stream" : "output larger than out_stride");
		}
	}
	HIP_TRY(hipStreamSynchronize(st));
	return first;
}

int cct_read_header(const uint8_t *h_file, size_t len, const char magic[4], cct_header *out)
{
	if (!h_file || !out) return fail(CCT_E_ARG, "null argument");
	if (len < 13) return fail(CCT_E_STREAM, "file shorter than the 13-byte header");
	if (memcmp(h_file, magic, 4) != 0) return fail(CCT_E_MAGIC, "Image does not contain valid header");
	out->width = (h_file[4] << 8) | h_file[5];
	out->height = (h_file[6] << 8) | h_file[7];
	out->channels = h_file[8];
	out->bytes_per_channel = h_file[9];
	out->fractal = h_file[10] != 0;
	out->segmentation = h_file[11] != 0;
	out->deflate = h_file[12] != 0;
	return CCT_OK;
}

int cct_decode_payload_dev(const uint8_t *d_payload, size_t payload_stride, const uint32_t *d_payload_sizes, int n,
                           int width, int height, int block_size, int fractal, uint16_t *d_images, uint32_t *d_status)
{
	std::lock_guard<std::mutex> lkd(g_mu_dec[0]);  // the workspaces of decode slot 0
	std::lock_guard<std::mutex> lk(g_mu);
	ApiCall in_call;
	int rc = check_shape(n, width, height, block_size);
	if (rc) return rc;
	if ((rc = ensure_ctx())) return rc;
	return decode_payload_locked(g_dec[0], d_payload, payload_stride, d_payload_sizes, n, width, height, block_size, fractal,
	                             d_images, d_status, g_ctx.stream);
}

int cct_decode_batch(const uint8_t *h_files, const uint64_t *h_offsets, int n, int block_size, const char magic[4],
                     uint16_t *images, int images_on_device, size_t images_cap_px, uint32_t *h_status)
{
	if (n < 0) return fail(CCT_E_ARG, "negative batch size");
	if (n == 0) return CCT_OK;
	if (!(g_ctx.ready && g_ctx.pid == getpid())) {  // first use in this process: bind the device before slot and shared lock (see above)
		std::lock_guard<std::mutex> lk(g_mu);
		int rc0 = ensure_ctx();
		if (rc0) return rc0;
	}
	std::unique_lock<std::mutex> lkd;
	DecSlot &D = acquire_decode_slot(lkd);
	ApiCall in_call;
	cct_header h0;
	int rc = cct_read_header(h_files + h_offsets[0], (size_t)(h_offsets[1] - h_offsets[0]), magic, &h0);
	if (rc) return rc;
	for (int i = 1; i < n; i++) {
		cct_header h;
		if ((rc = cct_read_header(h_files + h_offsets[i], (size_t)(h_offsets[i + 1] - h_offsets[i]), magic, &h))) return rc;
		if (h.width != h0.width || h.height != h0.height || h.fractal != h0.fractal || h.deflate != h0.deflate)
			return fail(CCT_E_MIXED, "file %d differs from file 0 in shape or flags", i);
	}
	if (h0.width == 0 || h0.height == 0) return fail(CCT_E_SHAPE, "empty image");
	if ((rc = check_shape(n, h0.width, h0.height, block_size))) return rc;
	const size_t N = (size_t)h0.width * h0.height;
	if (images_cap_px < (size_t)n * N) return fail(CCT_E_CAP, "output holds %zu pixels, need %zu", images_cap_px, (size_t)n * N);
	const size_t stride = cct_payload_stride(h0.width, h0.height, block_size);
	int zthreads = 1;
	{  // the slot's own buffers; no device lock, encodes and the other decode slot may be in flight
		HIP_TRY(hipSetDevice(g_ctx.device));
		if ((rc = D.dh_stage.ensure((size_t)n * stride))) return rc;
		if ((rc = D.d_payload.ensure((size_t)n * stride))) return rc;
		if ((rc = D.d_sizes.ensure((size_t)n * 4))) return rc;
		if ((rc = D.d_status.ensure((size_t)n * 4))) return rc;
		if (!images_on_device && (rc = D.d_images.ensure((size_t)n * N * 2))) return rc;
		zthreads = g_ctx.zlib_threads;
	}
	uint8_t *stage = (uint8_t *)D.dh_stage.p;
	std::vector<uint32_t> psz(n);
	for (int i = 0; i < n; i++) h_status[i] = CCT_OK;
	const bool dev_inflate = h0.deflate && g_ctx.device_inflate;
	std::vector<uint32_t> dst(n), zst(n, 0);
	std::vector<uint64_t> rel(dev_inflate ? n + 1 : 0);
	DrainOnExit drain(D.stream);  // copies into the vectors above / the caller's images must land before any return
	if (dev_inflate) {
		// INFLATE on the device (inflate_kernels.hip): the archive goes up as it is, payloads never touch the host
		const uint64_t a0 = h_offsets[0], a1 = h_offsets[n];
		const size_t abytes = (size_t)(a1 - a0), apad = (abytes + 31) & ~(size_t)15;
		HIP_TRY(hipSetDevice(g_ctx.device));
		if ((rc = D.d_arch.ensure(apad + 16))) return rc;
		if ((rc = D.d_archoffs.ensure((size_t)(n + 1) * 8))) return rc;
		if ((rc = D.d_zstatus.ensure((size_t)n * 4))) return rc;
		for (int i = 0; i <= n; i++) rel[i] = h_offsets[i] - a0;
		hipStream_t st = D.stream;
		const double t_inf0 = now_ms();
		HIP_TRY(hipMemcpyAsync(D.d_arch.p, h_files + a0, abytes, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(D.d_archoffs.p, rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
		InflateArgs ia{};
		ia.in = (const uint8_t *)D.d_arch.p; ia.in_total = apad;
		ia.offsets = (const uint64_t *)D.d_archoffs.p; ia.skip = 13;
		ia.out = (uint8_t *)D.d_payload.p; ia.out_stride = stride;
		ia.out_sizes = (uint32_t *)D.d_sizes.p; ia.status = (uint32_t *)D.d_zstatus.p;
		// A decode issued while an encode batch is on the device is part of a pipeline (the bench: the decode of step k next to
		// the encode of step k+1, with slack): its kernels are released when the next transform+pack stage has ended, so that
		// they run next to that batch's sort and match kernels and not next to a tree or transform+pack kernel
		// (sched_kernels.hip has the measurements).  The archive upload above does not wait.  Option "decode_yields" = 0
		// launches at once.
		if (g_ctx.decode_yields && g_encodes_in_flight.load(std::memory_order_relaxed) > 0)
			HIP_TRY(launch_gate_wait(g_gate, g_gate_passes_issued.load(std::memory_order_relaxed), 600u, 20000u, st));
		HIP_TRY(hipEventRecord(D.ev_d0, st));
		HIP_TRY(launch_inflate(ia, n, st, inflate_lanes_now()));
		HIP_TRY(hipEventRecord(D.ev_d1, st));
		uint16_t *d_img = images_on_device ? images : (uint16_t *)D.d_images.p;
		hipEvent_t ev_k = D.ev_k_dec0, ev_k1 = D.ev_k_dec1;
		HIP_TRY(hipEventRecord(ev_k, st));
		rc = decode_payload_locked(D, (const uint8_t *)D.d_payload.p, stride, (const uint32_t *)D.d_sizes.p, n, h0.width,
		                           h0.height, block_size, h0.fractal, d_img, (uint32_t *)D.d_status.p, st);
		if (rc) return rc;
		HIP_TRY(hipEventRecord(ev_k1, st));
		HIP_TRY(hipMemcpyAsync(dst.data(), D.d_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(zst.data(), D.d_zstatus.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
		if (!images_on_device)
			HIP_TRY(hipMemcpyAsync(images, d_img, (size_t)n * N * 2, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipEventElapsedTime(&tl_inflate_ms, D.ev_d0, D.ev_d1));
		HIP_TRY(hipEventElapsedTime(&tl_dec_kernel_ms, ev_k, ev_k1));
		(void)t_inf0;
	} else {
	// INFLATE stage: zlib.decompress(file_bytes[13:]), core.py:421 -- host threads, no device lock held
	const double t_inf0 = now_ms();
	static std::mutex mu_team;  // one host team for the decode side: two slots take turns on this path
	std::unique_lock<std::mutex> lk_team(mu_team);
	g_team_dec.run(n, h0.deflate ? zthreads : 1, [&](int i) {
		const uint8_t *body = h_files + h_offsets[i] + 13;
		const size_t blen = (size_t)(h_offsets[i + 1] - h_offsets[i]) - 13;
		uint8_t *dst = stage + (size_t)i * stride;
		if (h0.deflate) {
			uLongf dl = (uLongf)stride;
			const int zr = uncompress(dst, &dl, body, (uLong)blen);
			if (zr == Z_BUF_ERROR) { h_status[i] = CCT_E_STREAM; psz[i] = 0; }  // longer than any valid stream
			else if (zr != Z_OK) { h_status[i] = CCT_E_ZLIB; psz[i] = 0; }
			else psz[i] = (uint32_t)dl;
		} else {
			if (blen > stride) { h_status[i] = CCT_E_STREAM; psz[i] = 0; }
			else { memcpy(dst, body, blen); psz[i] = (uint32_t)blen; }
		}
	});
	lk_team.unlock();
	const float t_inflate = (float)(now_ms() - t_inf0);
	{  // device phase on the decode stream (no device lock: the HIP runtime is thread-safe)
		hipStream_t st = D.stream;
		tl_inflate_ms = t_inflate;
		// one strided copy of the used part of every staged payload (a copy per slice costs more in launches
		// than in bytes)
		size_t used = 16;
		for (int i = 0; i < n; i++) used = std::max(used, (size_t)((psz[i] + 15u) & ~15u));
		used = std::min(used, stride);
		const double t_h0 = now_ms();
		HIP_TRY(hipMemcpy2DAsync(D.d_payload.p, stride, stage, stride, used, (size_t)n, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(D.d_sizes.p, psz.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
		uint16_t *d_img = images_on_device ? images : (uint16_t *)D.d_images.p;
		HIP_TRY(hipEventRecord(D.ev_d0, st));
		rc = decode_payload_locked(D, (const uint8_t *)D.d_payload.p, stride, (const uint32_t *)D.d_sizes.p, n, h0.width,
		                           h0.height, block_size, h0.fractal, d_img, (uint32_t *)D.d_status.p, st);
		if (rc) return rc;
		HIP_TRY(hipEventRecord(D.ev_d1, st));
		HIP_TRY(hipMemcpyAsync(dst.data(), D.d_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
		if (!images_on_device)
			HIP_TRY(hipMemcpyAsync(images, d_img, (size_t)n * N * 2, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipEventElapsedTime(&tl_dec_kernel_ms, D.ev_d0, D.ev_d1));
		if (getenv("CCT_TRACE"))
			fprintf(stderr, "[cct] decode n=%d: inflate %.2f ms, h2d+kernel+sync %.2f ms (kernel %.2f), used %zu of stride %zu\n", n,
			        t_inflate, now_ms() - t_h0, tl_dec_kernel_ms, used, stride);
	}
	}
	int first = CCT_OK;
	for (int i = 0; i < n; i++) {
		if (h_status[i] == CCT_OK) {
			if (zst[i] & CCT_ST_ZLIB) h_status[i] = CCT_E_ZLIB;
			else if ((zst[i] | dst[i]) & CCT_ST_STREAM) h_status[i] = CCT_E_STREAM;
			else if (dst[i] & CCT_ST_OVERFLOW) h_status[i] = CCT_E_OVERFLOW;
		}
		if (h_status[i] != CCT_OK && first == CCT_OK) {
			first = (int)h_status[i];
			fail(first, "file %d: %s", i,
			     first == CCT_E_ZLIB ? "invalid DEFLATE stream"
			     : first == CCT_E_OVERFLOW ? "int too big to convert (pixel left the 16-bit range)"
			                               : "truncated or malformed token stream");
		}
	}
	return first;
}

int cct_last_timings(float *out6)
{
	// no lock: this only reads a few floats, and taking the device lock here would park the caller behind a
	// whole batch that another thread is encoding
	out6[0] = tl_enc_kernel_ms; out6[1] = tl_d2h_ms; out6[2] = tl_deflate_ms;
	out6[3] = tl_inflate_ms; out6[4] = tl_dec_kernel_ms; out6[5] = tl_h2d_ms;
	return CCT_OK;
}

int cct_set_option(const char *key, int value)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!strcmp(key, "zlib_threads")) { if (value < 1) return fail(CCT_E_ARG, "zlib_threads < 1"); g_ctx.zlib_threads = value; return CCT_OK; }
	if (!strcmp(key, "tile_path")) { g_ctx.use_tiles = (value >= 0 && value <= 4) ? value : 1; return CCT_OK; }
	if (!strcmp(key, "stream_tpg")) { g_ctx.stream_tpg = (value == 1 || value == 2 || value == 4) ? value : STREAM_TPG; return CCT_OK; }
	if (!strcmp(key, "debug_skip")) { g_ctx.dbg_skip = value; return CCT_OK; }
	if (!strcmp(key, "pipe_tpw")) { g_ctx.pipe_tpw = value; return CCT_OK; }
	if (!strcmp(key, "pipe_timing")) { g_ctx.pipe_timing = value; return CCT_OK; }
	if (!strcmp(key, "device_deflate")) { g_ctx.device_deflate = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "device_inflate")) { g_ctx.device_inflate = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "deflate_graph")) { g_ctx.use_graph = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "deflate_compact_records")) { g_ctx.compact_recs = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "decode_yields")) { g_ctx.decode_yields = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "queue_ahead")) { g_ctx.queue_ahead = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "deflate_fork")) { g_ctx.deflate_fork = value ? 1 : 0; return CCT_OK; }
	if (!strcmp(key, "encode_slots")) { g_ctx.enc_slots = std::max(1, std::min(value, N_ENC_SLOTS)); return CCT_OK; }
	if (!strcmp(key, "decode_slots")) { g_ctx.dec_slots = std::max(1, std::min(value, N_DEC_SLOTS)); return CCT_OK; }
	if (!strcmp(key, "inflate_lanes")) {
		if (value != 0 && value != 256 && value != 512) return fail(CCT_E_ARG, "inflate_lanes must be 0 (automatic), 256 or 512");
		g_ctx.inflate_lanes = value; return CCT_OK;
	}
	if (!strcmp(key, "wg_threads")) {
		if (value != 256 && value != 512 && value != 1024) return fail(CCT_E_ARG, "wg_threads must be 256, 512 or 1024");
		g_ctx.wg_threads = value; return CCT_OK;
	}
	return fail(CCT_E_ARG, "unknown option %s", key);
}
int cct_get_option(const char *key, int *value)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!strcmp(key, "zlib_threads")) { *value = g_ctx.zlib_threads; return CCT_OK; }
	if (!strcmp(key, "tile_path")) { *value = g_ctx.use_tiles; return CCT_OK; }
	if (!strcmp(key, "stream_tpg")) { *value = g_ctx.stream_tpg; return CCT_OK; }
	if (!strcmp(key, "last_encode_path")) { *value = g_ctx.last_path; return CCT_OK; }
	if (!strncmp(key, "pipe_us_k", 9) && key[9] >= '1' && key[9] <= '4') { *value = (int)(g_ctx.pipe_us[key[9] - '1'] * 10.0f); return CCT_OK; }
	if (!strcmp(key, "device_deflate")) { *value = g_ctx.device_deflate; return CCT_OK; }
	if (!strcmp(key, "device_inflate")) { *value = g_ctx.device_inflate; return CCT_OK; }
	if (!strcmp(key, "deflate_graph")) { *value = g_ctx.use_graph; return CCT_OK; }
	if (!strcmp(key, "deflate_compact_records")) { *value = g_ctx.compact_recs; return CCT_OK; }
	if (!strcmp(key, "decode_yields")) { *value = g_ctx.decode_yields; return CCT_OK; }
	if (!strcmp(key, "queue_ahead")) { *value = g_ctx.queue_ahead; return CCT_OK; }
	if (!strcmp(key, "deflate_fork")) { *value = g_ctx.deflate_fork; return CCT_OK; }
	if (!strcmp(key, "encode_slots")) { *value = g_ctx.enc_slots; return CCT_OK; }
	if (!strcmp(key, "decode_slots")) { *value = g_ctx.dec_slots; return CCT_OK; }
	if (!strcmp(key, "inflate_lanes")) { *value = g_ctx.inflate_lanes; return CCT_OK; }
	if (!strcmp(key, "last_inflate_lanes")) { *value = g_ctx.last_inflate_lanes; return CCT_OK; }
	if (!strcmp(key, "wg_threads")) { *value = g_ctx.wg_threads; return CCT_OK; }
	return fail(CCT_E_ARG, "unknown option %s", key);
}

}  // extern "C"
