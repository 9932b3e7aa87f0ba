// Host generator of the traversal table: the generalized Hilbert ("gilbert") curve that
// GeneralizedHilbertCurve(width, height, get_index=True).generate_all() produces
// (reference src/codec/curve.py:45-138).  The table depends on the shape only, so the library
// builds it once per (width, height), keeps it in HBM and every kernel indexes it.
//
// Written as an explicit work stack (no recursion, no generators): each frame is one
// sub-rectangle given by an origin and two axis vectors; leaves are straight runs.
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace cct {

namespace {
struct Frame { int64_t x, y, ax, ay, bx, by; };
inline int64_t sgn(int64_t v) { return (v > 0) - (v < 0); }
inline int64_t iabs(int64_t v) { return v < 0 ? -v : v; }
inline int64_t half_floor(int64_t v) { return v >> 1; }  // Python's `//2` (floor), curve.py:115-116
}  // namespace

// width = image.shape[0], height = image.shape[1]; emits y*width + x (curve.py:71-74).
bool gilbert_table(int width, int height, int32_t *out)
{
	const int64_t total = (int64_t)width * height;
	if (width <= 0 || height <= 0) return total == 0;
	std::vector<Frame> stack;
	stack.reserve(256);
	if (width >= height) stack.push_back({0, 0, width, 0, 0, height});  // curve.py:66-67
	else stack.push_back({0, 0, 0, height, width, 0});                  // curve.py:68-69
	int64_t n = 0;
	while (!stack.empty()) {
		const Frame f = stack.back();
		stack.pop_back();
		const int64_t w = iabs(f.ax + f.ay), h = iabs(f.bx + f.by);
		const int64_t dax = sgn(f.ax), day = sgn(f.ay), dbx = sgn(f.bx), dby = sgn(f.by);
		if (h == 1 || w == 1) {  // straight run along the major (h == 1) or the minor axis
			const int64_t len = (h == 1) ? w : h;
			const int64_t sx = (h == 1) ? dax : dbx, sy = (h == 1) ? day : dby;
			int64_t x = f.x, y = f.y;
			if (n + len > total) return false;
			for (int64_t i = 0; i < len; i++, x += sx, y += sy) out[n++] = (int32_t)(y * width + x);
			continue;
		}
		int64_t ax2 = half_floor(f.ax), ay2 = half_floor(f.ay);
		int64_t bx2 = half_floor(f.bx), by2 = half_floor(f.by);
		const int64_t w2 = iabs(ax2 + ay2), h2 = iabs(bx2 + by2);
		if (2 * w > 3 * h) {  // wide: split the major axis in two, curve.py:121-128
			if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }
			// children are pushed in reverse so they pop in traversal order
			stack.push_back({f.x + ax2, f.y + ay2, f.ax - ax2, f.ay - ay2, f.bx, f.by});
			stack.push_back({f.x, f.y, ax2, ay2, f.bx, f.by});
		} else {              // up, across, down, curve.py:130-138
			if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
			stack.push_back({f.x + (f.ax - dax) + (bx2 - dbx), f.y + (f.ay - day) + (by2 - dby),
			                 -bx2, -by2, -(f.ax - ax2), -(f.ay - ay2)});
			stack.push_back({f.x + bx2, f.y + by2, f.ax, f.ay, f.bx - bx2, f.by - by2});
			stack.push_back({f.x, f.y, bx2, by2, ax2, ay2});
		}
	}
	return n == total;
}

}  // namespace cct
