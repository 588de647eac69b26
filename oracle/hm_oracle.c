/*
 * hm_oracle.c -- CPU restatement of HM 16.4's integer motion-estimation arithmetic.
 * TEST INFRASTRUCTURE ONLY (see hm_oracle.h).  Parity pinned against oracle/_ref and
 * tests/golden/.  Citations are relative to /root/reference/source/Lib/.
 */
#include "hm_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* MV-bit cost                                                                          */
/* ------------------------------------------------------------------------------------ */

/* TComRdCost::xGetComponentBits, TLibCommon/TComRdCost.cpp:278-292 (same loop as
 * cl/sad.cl:374-396): exp-Golomb length of a signed MV difference. */
uint32_t hmo_component_bits(int val) {
  uint32_t length = 1;
  uint32_t temp = (val <= 0) ? ((uint32_t)(-val) << 1) + 1 : ((uint32_t)val << 1);
  while (temp != 1) {
    temp >>= 1;
    length += 2;
  }
  return length;
}

/* TComRdCost::setLambda, TComRdCost.cpp:209: m_uiLambdaMotionSAD = (UInt)floor(65536*sqrt(l));
 * TEncOpenCL::setLambda (TLibEncoder/TEncOpenCL.h:121) uses the identical expression. */
uint32_t hmo_lambda_q16(double lambda) { return (uint32_t)floor(65536.0 * sqrt(lambda)); }

/* TComRdCost::getCost(x,y) / getBits, TComRdCost.h:172-189:
 *   m_uiCost * (bits((x<<scale)-pred.x) + bits((y<<scale)-pred.y)) >> 16   in UInt (wraps). */
uint32_t hmo_mv_cost(uint32_t lambda_q16, int x, int y, int pred_x, int pred_y, int scale) {
  uint32_t bits = hmo_component_bits(x * (1 << scale) - pred_x) + hmo_component_bits(y * (1 << scale) - pred_y);
  return (uint32_t)(lambda_q16 * bits) >> 16;
}

/* ------------------------------------------------------------------------------------ */
/* SAD                                                                                  */
/* ------------------------------------------------------------------------------------ */

/* TComRdCost::xGetSAD4/8/16/32/64/12/24/48, TComRdCost.cpp:493-964: every variant is
 *   for (rows; step 1<<sub) sum += |org[n]-cur[n]| for n<cols;  sum <<= sub;
 *   return sum >> DISTORTION_PRECISION_ADJUSTMENT(bitDepth-8)     (TypeDef.h:280-284: = x) */
uint32_t hmo_sad(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride,
                 int w, int h, int sub_shift, int bit_depth) {
  const int step = 1 << sub_shift;
  uint32_t sum = 0;
  for (int rows = h; rows != 0; rows -= step) {
    for (int n = 0; n < w; ++n) sum += (uint32_t)abs((int)org[n] - (int)cur[n]);
    org += org_stride * step;
    cur += cur_stride * step;
  }
  sum <<= sub_shift;
  return sum >> (bit_depth - 8);
}

/* TComRdCostWeightPrediction::xGetSADw, TComRdCostWeightPrediction.cpp:55-90:
 *   pred = ((w0 * cur[n] + round) >> shift) + offset        (:79, not clipped)
 *   sum += |org[n] - pred| over EVERY row; return sum >> (bitDepth-8)   (:81, :89) */
static inline int wp_pred(int v, const hmo_wp* wp) {
  return (hmo_pel)(((wp->w0 * v + wp->round) >> wp->shift) + wp->offset);   /* `const Pel pred = ...` (:79): the value is kept in a Pel = Short */
}
uint32_t hmo_sad_w(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride, int w, int h, int bit_depth, const hmo_wp* wp) {
  uint32_t sum = 0;
  for (int rows = h; rows != 0; --rows) {
    for (int n = 0; n < w; ++n) sum += (uint32_t)abs((int)org[n] - wp_pred(cur[n], wp));
    org += org_stride;
    cur += cur_stride;
  }
  return sum >> (bit_depth - 8);
}

/* ------------------------------------------------------------------------------------ */
/* slot layout                                                                          */
/* ------------------------------------------------------------------------------------ */

/* Closed form of the 593-case switch in TComDataCU::getIndexBlock (TComDataCU.cpp:4676-6461);
 * same base offsets as the hard-coded ones in cl/sad.cl:200-365.  CU size S in {8,16,32,64},
 * n = 64/S CUs per row, CU raster index r = cy*n + cx. */
static const int k_base_2NxN[4] = {0, 448, 560, 588};
static const int k_base_Nx2N[4] = {128, 480, 568, 590};
static const int k_base_AMP[4] = {-1, 256, 512, 576};
static const int k_base_2Nx2N[4] = {384, 544, 584, 592};

static int size_level(int s) { return s == 8 ? 0 : s == 16 ? 1 : s == 32 ? 2 : s == 64 ? 3 : -1; }

/* PartSize enum values, TLibCommon/TypeDef.h:409-416 */
enum { P_2Nx2N = 0, P_2NxN = 1, P_Nx2N = 2, P_NxN = 3, P_2NxnU = 4, P_2NxnD = 5, P_nLx2N = 6, P_nRx2N = 7 };

static int slot_of(int part_size, int part_idx, int lvl, int cx, int cy) {
  const int n = 8 >> lvl; /* 64/S */
  const int r = cy * n + cx;
  switch (part_size) {
    case P_2Nx2N: return part_idx == 0 ? k_base_2Nx2N[lvl] + r : -1;
    case P_2NxN: return k_base_2NxN[lvl] + cy * 2 * n + part_idx * n + cx;
    case P_Nx2N: return k_base_Nx2N[lvl] + cy * 2 * n + 2 * cx + part_idx;
    default: break;
  }
  if (lvl == 0) return -1; /* no AMP for 8x8 CUs */
  int k;
  switch (part_size * 2 + part_idx) {
    case P_2NxnU * 2 + 0: k = 0; break; /* S x S/4  top    */
    case P_2NxnD * 2 + 1: k = 1; break; /* S x S/4  bottom */
    case P_2NxnD * 2 + 0: k = 2; break; /* S x 3S/4 top    */
    case P_2NxnU * 2 + 1: k = 3; break; /* S x 3S/4 bottom */
    case P_nLx2N * 2 + 0: k = 4; break; /* S/4 x S  left   */
    case P_nRx2N * 2 + 1: k = 5; break; /* S/4 x S  right  */
    case P_nRx2N * 2 + 0: k = 6; break; /* 3S/4 x S left   */
    case P_nLx2N * 2 + 1: k = 7; break; /* 3S/4 x S right  */
    default: return -1;
  }
  return k_base_AMP[lvl] + k * n * n + r;
}

/* rectangle of a PU inside its CU: TComDataCU::getPartIndexAndSize (TComDataCU.cpp) */
static void pu_rect(int part_size, int part_idx, int s, hmo_rect* r) {
  r->x = 0; r->y = 0; r->w = s; r->h = s;
  switch (part_size) {
    case P_2NxN: r->h = s / 2; r->y = part_idx ? s / 2 : 0; break;
    case P_Nx2N: r->w = s / 2; r->x = part_idx ? s / 2 : 0; break;
    case P_2NxnU: r->h = part_idx ? 3 * s / 4 : s / 4; r->y = part_idx ? s / 4 : 0; break;
    case P_2NxnD: r->h = part_idx ? s / 4 : 3 * s / 4; r->y = part_idx ? 3 * s / 4 : 0; break;
    case P_nLx2N: r->w = part_idx ? 3 * s / 4 : s / 4; r->x = part_idx ? s / 4 : 0; break;
    case P_nRx2N: r->w = part_idx ? s / 4 : 3 * s / 4; r->x = part_idx ? 3 * s / 4 : 0; break;
    default: break;
  }
}

static hmo_rect g_slot_rect[HMO_NUM_CTU_PARTS];
static int g_slot_init = 0;

static void init_slots(void) {
  if (g_slot_init) return;
  static const int part_sizes[7] = {P_2Nx2N, P_2NxN, P_Nx2N, P_2NxnU, P_2NxnD, P_nLx2N, P_nRx2N};
  for (int i = 0; i < HMO_NUM_CTU_PARTS; ++i) g_slot_rect[i].w = 0;
  for (int lvl = 0; lvl < 4; ++lvl) {
    const int s = 8 << lvl, n = 8 >> lvl;
    for (int cy = 0; cy < n; ++cy)
      for (int cx = 0; cx < n; ++cx)
        for (int pi = 0; pi < 7; ++pi)
          for (int part_idx = 0; part_idx < 2; ++part_idx) {
            int slot = slot_of(part_sizes[pi], part_idx, lvl, cx, cy);
            if (slot < 0) continue;
            hmo_rect r;
            pu_rect(part_sizes[pi], part_idx, s, &r);
            r.x += cx * s;
            r.y += cy * s;
            g_slot_rect[slot] = r;
          }
  }
  g_slot_init = 1;
}

int hmo_slot_rect(int slot, hmo_rect* r) {
  init_slots();
  if (slot < 0 || slot >= HMO_NUM_CTU_PARTS || g_slot_rect[slot].w == 0) return -1;
  *r = g_slot_rect[slot];
  return 0;
}

/* key built by TComDataCU::getIndexBlock, TComDataCU.cpp:3379-3391 */
int32_t hmo_index_key(int part_size, int depth, int part_idx, int abs_z_idx, int cu_h, int cu_w) {
  int32_t t = part_size + depth * 10 + part_idx * 100;
  t = abs_z_idx + t * 1000;
  t = cu_h + t * 100;
  t = cu_w + t * 100;
  return t;
}

/* z-order index (4x4 units, 16x16 grid) -> x,y in 4x4 units (g_auiZscanToRaster, TComRom.cpp) */
static void zidx_to_xy(int z, int* x, int* y) {
  int xx = 0, yy = 0;
  for (int b = 0; b < 4; ++b) {
    xx |= ((z >> (2 * b)) & 1) << b;
    yy |= ((z >> (2 * b + 1)) & 1) << b;
  }
  *x = xx; *y = yy;
}

int hmo_index_block(int part_size, int depth, int part_idx, int abs_z_idx, int cu_size) {
  const int lvl = size_level(cu_size);
  if (lvl < 0 || depth != 3 - lvl) return -1;
  int bx, by;
  zidx_to_xy(abs_z_idx, &bx, &by);
  if ((bx * 4) % cu_size || (by * 4) % cu_size) return -1;
  return slot_of(part_size, part_idx, lvl, bx * 4 / cu_size, by * 4 / cu_size);
}

/* ------------------------------------------------------------------------------------ */
/* search window                                                                        */
/* ------------------------------------------------------------------------------------ */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* TComDataCU::clipMv, TComDataCU.cpp:2907-2920 (quarter-pel units).  The TComMv fields are
 * Short, so the result is narrowed like TComMv::setHor does. */
void hmo_clip_mv(int* mvx_q, int* mvy_q, int cu_x, int cu_y, int pic_w, int pic_h, int max_cu) {
  const int shift = 2, offset = 8;
  /* `* (1 << shift)`: the reference shifts negative values left (`<< mvShift`), which C leaves undefined; same arithmetic */
  const int hor_max = (pic_w + offset - cu_x - 1) * (1 << shift);
  const int hor_min = (-max_cu - offset - cu_x + 1) * (1 << shift);
  const int ver_max = (pic_h + offset - cu_y - 1) * (1 << shift);
  const int ver_min = (-max_cu - offset - cu_y + 1) * (1 << shift);
  *mvx_q = (int16_t)imin(hor_max, imax(hor_min, *mvx_q));
  *mvy_q = (int16_t)imin(ver_max, imax(ver_min, *mvy_q));
}

/* TEncSearch::xSetSearchRange, TLibEncoder/TEncSearch.cpp:3814-3830 */
void hmo_set_search_range(int pred_x_q, int pred_y_q, int sr, int cu_x, int cu_y, int pic_w,
                          int pic_h, int max_cu, int* lt_x, int* lt_y, int* rb_x, int* rb_y) {
  int px = pred_x_q, py = pred_y_q;
  hmo_clip_mv(&px, &py, cu_x, cu_y, pic_w, pic_h, max_cu);
  int ltx = (int16_t)(px - (sr << 2)), lty = (int16_t)(py - (sr << 2));
  int rbx = (int16_t)(px + (sr << 2)), rby = (int16_t)(py + (sr << 2));
  hmo_clip_mv(&ltx, &lty, cu_x, cu_y, pic_w, pic_h, max_cu);
  hmo_clip_mv(&rbx, &rby, cu_x, cu_y, pic_w, pic_h, max_cu);
  *lt_x = ltx >> 2; *lt_y = lty >> 2; /* TComMv::operator>>= is an arithmetic shift */
  *rb_x = rbx >> 2; *rb_y = rby >> 2;
}

/* ------------------------------------------------------------------------------------ */
/* exhaustive search, one PU: TEncSearch::xPatternSearch, TEncSearch.cpp:3835-3897        */
/* ------------------------------------------------------------------------------------ */
void hmo_pattern_search(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref,
                        int ref_stride, const hmo_params* p, int* mvx, int* mvy, uint32_t* sad) {
  uint32_t best = UINT_MAX;
  int best_x = 0, best_y = 0;
  /* :3853-3859  fast encoder decision: sub-sampled SAD when the block has more than 8 rows */
  const int sub_shift = (p->fen && h > 8) ? 1 : 0;
  const hmo_pel* row = ref + (ptrdiff_t)p->lt_y * ref_stride;
  for (int y = p->lt_y; y <= p->rb_y; ++y) {
    for (int x = p->lt_x; x <= p->rb_x; ++x) {
      uint32_t s = hmo_sad(org, org_stride, row + x, ref_stride, w, h, sub_shift, p->bit_depth);
      s += hmo_mv_cost(p->lambda_q16, x, y, p->pred_x, p->pred_y, 2); /* :3880, cost scale 2 (:3738) */
      if (s < best) { /* strict '<': first candidate in raster order wins ties (:3882) */
        best = s; best_x = x; best_y = y;
      }
    }
    row += ref_stride;
  }
  *mvx = best_x; *mvy = best_y;
  *sad = best - hmo_mv_cost(p->lambda_q16, best_x, best_y, p->pred_x, p->pred_y, 2); /* :3895 */
}

/* xPatternSearch in a slice with weighted prediction: the DistFunc of every candidate is xGetSADw (TComRdCost.cpp:467-469 and the
 * same two lines at the head of each width's function), so iSubShift set at TEncSearch.cpp:3853-3859 has no effect */
void hmo_pattern_search_w(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, const hmo_params* p,
                          const hmo_wp* wp, int* mvx, int* mvy, uint32_t* sad) {
  uint32_t best = UINT_MAX;
  int best_x = 0, best_y = 0;
  const hmo_pel* row = ref + (ptrdiff_t)p->lt_y * ref_stride;
  for (int y = p->lt_y; y <= p->rb_y; ++y) {
    for (int x = p->lt_x; x <= p->rb_x; ++x) {
      uint32_t s = hmo_sad_w(org, org_stride, row + x, ref_stride, w, h, p->bit_depth, wp);
      s += hmo_mv_cost(p->lambda_q16, x, y, p->pred_x, p->pred_y, 2);
      if (s < best) { best = s; best_x = x; best_y = y; }
    }
    row += ref_stride;
  }
  *mvx = best_x; *mvy = best_y;
  *sad = best - hmo_mv_cost(p->lambda_q16, best_x, best_y, p->pred_x, p->pred_y, 2);
}

/* all 593 slots with weighted prediction: the weighted window is formed once (the prediction of a sample does not depend on the
 * candidate), then the unweighted all-slot search runs on it with every row counted */
void hmo_search_ctu_w(const hmo_pel* ctu, int ctu_stride, const hmo_pel* ref, int ref_stride, const hmo_params* p, const hmo_wp* wp,
                      int32_t* out_x, int32_t* out_y, uint32_t* out_sad, uint32_t* out_cost) {
  const int wx = p->rb_x - p->lt_x + 64, wy = p->rb_y - p->lt_y + 64;   /* samples the window spans */
  hmo_pel* win = (hmo_pel*)malloc(sizeof(hmo_pel) * (size_t)wx * wy);
  for (int y = 0; y < wy; ++y)
    for (int x = 0; x < wx; ++x) win[(size_t)y * wx + x] = (hmo_pel)wp_pred(ref[(ptrdiff_t)(p->lt_y + y) * ref_stride + p->lt_x + x], wp);
  hmo_params q = *p;
  q.fen = 0;
  hmo_search_ctu(ctu, ctu_stride, win - ((ptrdiff_t)p->lt_y * wx + p->lt_x), wx, &q, out_x, out_y, out_sad, out_cost);
  free(win);
}

void hmo_ocl_compat_params(hmo_params* p, int lt_x, int lt_y, int sr, uint32_t lambda_q16) {
  /* TEncOpenCL.cpp:312-313 scans x,y in [0, 2*sr] from LT; cl/sad.cl:374-398 uses pred (0,0),
   * all rows, no bit-depth shift. */
  p->lt_x = lt_x; p->lt_y = lt_y;
  p->rb_x = lt_x + 2 * sr; p->rb_y = lt_y + 2 * sr;
  p->pred_x = 0; p->pred_y = 0;
  p->lambda_q16 = lambda_q16;
  p->fen = 0; p->bit_depth = 8;
}

/* ------------------------------------------------------------------------------------ */
/* exhaustive search, all 593 slots of a CTU                                            */
/* ------------------------------------------------------------------------------------ */
void hmo_search_ctu(const hmo_pel* ctu, int ctu_stride, const hmo_pel* ref, int ref_stride,
                    const hmo_params* p, int32_t* out_x, int32_t* out_y, uint32_t* out_sad,
                    uint32_t* out_cost) {
  init_slots();
  uint32_t best[HMO_NUM_CTU_PARTS];
  for (int s = 0; s < HMO_NUM_CTU_PARTS; ++s) { best[s] = UINT_MAX; out_x[s] = 0; out_y[s] = 0; }
  /* integral images over the 16x16 grid of 4x4 blocks: ev = rows 0,2 of each block (what a
   * sub-sampled SAD of a 4-aligned rectangle reads), al = all 4 rows */
  uint32_t ev[17][17], al[17][17];
  memset(ev, 0, sizeof ev);
  memset(al, 0, sizeof al);
  for (int y = p->lt_y; y <= p->rb_y; ++y) {
    for (int x = p->lt_x; x <= p->rb_x; ++x) {
      const hmo_pel* c = ref + (ptrdiff_t)y * ref_stride + x;
      for (int by = 0; by < 16; ++by) {
        uint32_t row_ev = 0, row_al = 0;
        for (int bx = 0; bx < 16; ++bx) {
          uint32_t e = 0, o = 0;
          for (int r = 0; r < 4; ++r) {
            const hmo_pel* a = ctu + (by * 4 + r) * ctu_stride + bx * 4;
            const hmo_pel* b = c + (ptrdiff_t)(by * 4 + r) * ref_stride + bx * 4;
            uint32_t t = (uint32_t)(abs(a[0] - b[0]) + abs(a[1] - b[1]) + abs(a[2] - b[2]) + abs(a[3] - b[3]));
            if (r & 1) o += t; else e += t;
          }
          row_ev += e; row_al += e + o;
          ev[by + 1][bx + 1] = ev[by][bx + 1] + row_ev;
          al[by + 1][bx + 1] = al[by][bx + 1] + row_al;
        }
      }
      const uint32_t cost = hmo_mv_cost(p->lambda_q16, x, y, p->pred_x, p->pred_y, 2);
      for (int s = 0; s < HMO_NUM_CTU_PARTS; ++s) {
        const hmo_rect r = g_slot_rect[s];
        const int x0 = r.x >> 2, y0 = r.y >> 2, x1 = (r.x + r.w) >> 2, y1 = (r.y + r.h) >> 2;
        uint32_t sum;
        if (p->fen && r.h > 8) sum = (ev[y1][x1] - ev[y0][x1] - ev[y1][x0] + ev[y0][x0]) << 1;
        else sum = al[y1][x1] - al[y0][x1] - al[y1][x0] + al[y0][x0];
        sum >>= (p->bit_depth - 8);
        const uint32_t tot = sum + cost;
        if (tot < best[s]) { best[s] = tot; out_x[s] = x; out_y[s] = y; }
      }
    }
  }
  for (int s = 0; s < HMO_NUM_CTU_PARTS; ++s) {
    if (out_cost) out_cost[s] = best[s];
    out_sad[s] = best[s] - hmo_mv_cost(p->lambda_q16, out_x[s], out_y[s], p->pred_x, p->pred_y, 2);
  }
}

/* ------------------------------------------------------------------------------------ */
/* TZ search: TEncSearch.cpp:340-441 (help), :445-585 (2 point), :636-808 (diamond),     */
/*            :3935-4136 (driver), configuration :305-321                                */
/* ------------------------------------------------------------------------------------ */
typedef struct tz_state {
  const hmo_pel* org; int org_stride, w, h;
  const hmo_pel* ref; int ref_stride;
  const hmo_params* p;
  int sub_shift;
  uint32_t best_sad; int best_x, best_y;
  uint32_t best_dist, best_round; unsigned char point_nr;
  long probes;
} tz_state;

/* xTZSearchHelp :340-441 (non-SELECTIVE branch) */
static void tz_help(tz_state* t, int sx, int sy, unsigned char point_nr, uint32_t dist) {
  const hmo_pel* c = t->ref + (ptrdiff_t)sy * t->ref_stride + sx;
  uint32_t s = hmo_sad(t->org, t->org_stride, c, t->ref_stride, t->w, t->h, t->sub_shift, t->p->bit_depth);
  s += hmo_mv_cost(t->p->lambda_q16, sx, sy, t->p->pred_x, t->p->pred_y, 2);
  t->probes++;
  if (s < t->best_sad) {
    t->best_sad = s; t->best_x = sx; t->best_y = sy;
    t->best_dist = dist; t->best_round = 0; t->point_nr = point_nr;
  }
}

/* xTZ2PointSearch :445-585 */
static void tz_2point(tz_state* t, int l, int r, int tp, int b) {
  const int sx = t->best_x, sy = t->best_y;
  switch (t->point_nr) {
    case 1:
      if (sx - 1 >= l) tz_help(t, sx - 1, sy, 0, 2);
      if (sy - 1 >= tp) tz_help(t, sx, sy - 1, 0, 2);
      break;
    case 2:
      if (sy - 1 >= tp) {
        if (sx - 1 >= l) tz_help(t, sx - 1, sy - 1, 0, 2);
        if (sx + 1 <= r) tz_help(t, sx + 1, sy - 1, 0, 2);
      }
      break;
    case 3:
      if (sy - 1 >= tp) tz_help(t, sx, sy - 1, 0, 2);
      if (sx + 1 <= r) tz_help(t, sx + 1, sy, 0, 2);
      break;
    case 4:
      if (sx - 1 >= l) {
        if (sy + 1 <= b) tz_help(t, sx - 1, sy + 1, 0, 2);
        if (sy - 1 >= tp) tz_help(t, sx - 1, sy - 1, 0, 2);
      }
      break;
    case 5:
      if (sx + 1 <= r) {
        if (sy - 1 >= tp) tz_help(t, sx + 1, sy - 1, 0, 2);
        if (sy + 1 <= b) tz_help(t, sx + 1, sy + 1, 0, 2);
      }
      break;
    case 6:
      if (sx - 1 >= l) tz_help(t, sx - 1, sy, 0, 2);
      if (sy + 1 <= b) tz_help(t, sx, sy + 1, 0, 2);
      break;
    case 7:
      if (sy + 1 <= b) {
        if (sx - 1 >= l) tz_help(t, sx - 1, sy + 1, 0, 2);
        if (sx + 1 <= r) tz_help(t, sx + 1, sy + 1, 0, 2);
      }
      break;
    case 8:
      if (sx + 1 <= r) tz_help(t, sx + 1, sy, 0, 2);
      if (sy + 1 <= b) tz_help(t, sx, sy + 1, 0, 2);
      break;
    default: break;
  }
}

/* xTZ8PointDiamondSearch :636-808 */
static void tz_diamond(tz_state* t, int l, int r, int tp, int b, int sx, int sy, int dist) {
  const int top = sy - dist, bot = sy + dist, left = sx - dist, right = sx + dist;
  t->best_round += 1;
  if (dist == 1) {
    if (top >= tp) tz_help(t, sx, top, 2, dist);
    if (left >= l) tz_help(t, left, sy, 4, dist);
    if (right <= r) tz_help(t, right, sy, 5, dist);
    if (bot <= b) tz_help(t, sx, bot, 7, dist);
  } else if (dist <= 8) {
    const int top2 = sy - (dist >> 1), bot2 = sy + (dist >> 1);
    const int left2 = sx - (dist >> 1), right2 = sx + (dist >> 1);
    if (top >= tp && left >= l && right <= r && bot <= b) {
      tz_help(t, sx, top, 2, dist);
      tz_help(t, left2, top2, 1, dist >> 1);
      tz_help(t, right2, top2, 3, dist >> 1);
      tz_help(t, left, sy, 4, dist);
      tz_help(t, right, sy, 5, dist);
      tz_help(t, left2, bot2, 6, dist >> 1);
      tz_help(t, right2, bot2, 8, dist >> 1);
      tz_help(t, sx, bot, 7, dist);
    } else {
      if (top >= tp) tz_help(t, sx, top, 2, dist);
      if (top2 >= tp) {
        if (left2 >= l) tz_help(t, left2, top2, 1, dist >> 1);
        if (right2 <= r) tz_help(t, right2, top2, 3, dist >> 1);
      }
      if (left >= l) tz_help(t, left, sy, 4, dist);
      if (right <= r) tz_help(t, right, sy, 5, dist);
      if (bot2 <= b) {
        if (left2 >= l) tz_help(t, left2, bot2, 6, dist >> 1);
        if (right2 <= r) tz_help(t, right2, bot2, 8, dist >> 1);
      }
      if (bot <= b) tz_help(t, sx, bot, 7, dist);
    }
  } else {
    if (top >= tp && left >= l && right <= r && bot <= b) {
      tz_help(t, sx, top, 0, dist);
      tz_help(t, left, sy, 0, dist);
      tz_help(t, right, sy, 0, dist);
      tz_help(t, sx, bot, 0, dist);
      for (int i = 1; i < 4; ++i) {
        const int yt = top + (dist >> 2) * i, yb = bot - (dist >> 2) * i;
        const int xl = sx - (dist >> 2) * i, xr = sx + (dist >> 2) * i;
        tz_help(t, xl, yt, 0, dist);
        tz_help(t, xr, yt, 0, dist);
        tz_help(t, xl, yb, 0, dist);
        tz_help(t, xr, yb, 0, dist);
      }
    } else {
      if (top >= tp) tz_help(t, sx, top, 0, dist);
      if (left >= l) tz_help(t, left, sy, 0, dist);
      if (right <= r) tz_help(t, right, sy, 0, dist);
      if (bot <= b) tz_help(t, sx, bot, 0, dist);
      for (int i = 1; i < 4; ++i) {
        const int yt = top + (dist >> 2) * i, yb = bot - (dist >> 2) * i;
        const int xl = sx - (dist >> 2) * i, xr = sx + (dist >> 2) * i;
        if (yt >= tp) {
          if (xl >= l) tz_help(t, xl, yt, 0, dist);
          if (xr <= r) tz_help(t, xr, yt, 0, dist);
        }
        if (yb <= b) {
          if (xl >= l) tz_help(t, xl, yb, 0, dist);
          if (xr <= r) tz_help(t, xr, yb, 0, dist);
        }
      }
    }
  }
}

/* xTZSearch :3935-4136 with TZ_SEARCH_CONFIGURATION :305-321 (iRaster 5, zero-vector test,
 * diamond first search with stop after 3 rounds [FASTME_SMOOTHER_MV=1, CommonDef.h:205],
 * raster search when best distance > 5, star refinement with diamond). */
long hmo_tz_search(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref,
                   int ref_stride, const hmo_params* p, const hmo_tz_ctx* tz,
                   const int* int_mv_2nx2n, int start_x_q, int start_y_q, int* mvx, int* mvy,
                   uint32_t* sad) {
  enum { RASTER = 5, FIRST_ROUNDS = 3 };
  /* diamond / 2-point searches always use the caller's window (pcMvSrchRngLT/RB) ... */
  const int L = p->lt_x, R = p->rb_x, T = p->lt_y, B = p->rb_y;
  /* ... the raster scan uses these locals, which the 2Nx2N-predictor branch re-centres */
  int rl = L, rr = R, rt = T, rb = B;
  tz_state t;
  memset(&t, 0, sizeof t);
  t.org = org; t.org_stride = org_stride; t.w = w; t.h = h;
  t.ref = ref; t.ref_stride = ref_stride; t.p = p;
  t.sub_shift = (p->fen && h > 8) ? 1 : 0; /* :354-360 */
  t.best_sad = UINT_MAX;

  int sx = start_x_q, sy = start_y_q; /* rcMv = *pcMvPred (TEncSearch.cpp:3778) */
  hmo_clip_mv(&sx, &sy, tz->cu_x, tz->cu_y, tz->pic_w, tz->pic_h, tz->max_cu);
  sx >>= 2; sy >>= 2;
  tz_help(&t, sx, sy, 0, 0);      /* median predictor */
  tz_help(&t, 0, 0, 0, 0);        /* bTestZeroVector */
  if (int_mv_2nx2n) {
    int ix = (int16_t)(int_mv_2nx2n[0] * 4), iy = (int16_t)(int_mv_2nx2n[1] * 4);
    hmo_clip_mv(&ix, &iy, tz->cu_x, tz->cu_y, tz->pic_w, tz->pic_h, tz->max_cu);
    ix >>= 2; iy >>= 2;
    tz_help(&t, ix, iy, 0, 0);
    hmo_set_search_range((int16_t)(t.best_x * 4), (int16_t)(t.best_y * 4), tz->sr, tz->cu_x, tz->cu_y,
                         tz->pic_w, tz->pic_h, tz->max_cu, &rl, &rt, &rr, &rb);
  }
  int start_x = t.best_x, start_y = t.best_y;
  for (int dist = 1; dist <= tz->sr; dist *= 2) { /* first search */
    tz_diamond(&t, L, R, T, B, start_x, start_y, dist);
    if (t.best_round >= FIRST_ROUNDS) break;
  }
  if (t.best_dist == 1) {
    t.best_dist = 0;
    tz_2point(&t, L, R, T, B);
  }
  if ((int)t.best_dist > RASTER) { /* raster search */
    t.best_dist = RASTER;
    for (int y = rt; y <= rb; y += RASTER)
      for (int x = rl; x <= rr; x += RASTER) tz_help(&t, x, y, 0, RASTER);
  }
  while (t.best_dist > 0) { /* star refinement */
    start_x = t.best_x; start_y = t.best_y;
    t.best_dist = 0; t.point_nr = 0;
    for (int dist = 1; dist < tz->sr + 1; dist *= 2) tz_diamond(&t, L, R, T, B, start_x, start_y, dist);
    if (t.best_dist == 1) {
      t.best_dist = 0;
      if (t.point_nr != 0) tz_2point(&t, L, R, T, B);
    }
  }
  *mvx = t.best_x; *mvy = t.best_y;
  *sad = t.best_sad - hmo_mv_cost(p->lambda_q16, t.best_x, t.best_y, p->pred_x, p->pred_y, 2);
  return t.probes;
}

/* ------------------------------------------------------------------------------------ */
/* frame helpers                                                                        */
/* ------------------------------------------------------------------------------------ */

/* TComPicYuv::extendPicBorder, TLibCommon/TComPicYuv.cpp:214-262 */
void hmo_extend_border(hmo_pel* pic, int stride, int w, int h, int mx, int my) {
  hmo_pel* pi = pic;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < mx; ++x) { pi[-mx + x] = pi[0]; pi[w + x] = pi[w - 1]; }
    pi += stride;
  }
  pi -= (stride + mx);
  for (int y = 0; y < my; ++y) memcpy(pi + (y + 1) * stride, pi, sizeof(hmo_pel) * (size_t)(w + 2 * mx));
  pi -= (ptrdiff_t)(h - 1) * stride;
  for (int y = 0; y < my; ++y) memcpy(pi - (y + 1) * stride, pi, sizeof(hmo_pel) * (size_t)(w + 2 * mx));
}

typedef struct frame_job {
  const hmo_pel* cur; int cur_stride; const hmo_pel* ref; int ref_stride;
  int pic_w, pic_h, sr; const int16_t* pred_q; uint32_t lambda_q16; int fen, bit_depth;
  int ctu_first, ctu_count, tid, n_threads;
  int32_t *out_x, *out_y; uint32_t* out_sad;
} frame_job;

static void* frame_worker(void* arg) {
  frame_job* j = (frame_job*)arg;
  const int ctus_x = (j->pic_w + HMO_CTU - 1) / HMO_CTU;
  for (int i = j->tid; i < j->ctu_count; i += j->n_threads) {
    const int ctu = j->ctu_first + i;
    const int cu_x = (ctu % ctus_x) * HMO_CTU, cu_y = (ctu / ctus_x) * HMO_CTU;
    hmo_params p;
    p.pred_x = j->pred_q ? j->pred_q[2 * ctu] : 0;
    p.pred_y = j->pred_q ? j->pred_q[2 * ctu + 1] : 0;
    p.lambda_q16 = j->lambda_q16; p.fen = j->fen; p.bit_depth = j->bit_depth;
    hmo_set_search_range(p.pred_x, p.pred_y, j->sr, cu_x, cu_y, j->pic_w, j->pic_h, HMO_CTU,
                         &p.lt_x, &p.lt_y, &p.rb_x, &p.rb_y);
    hmo_search_ctu(j->cur + (ptrdiff_t)cu_y * j->cur_stride + cu_x, j->cur_stride,
                   j->ref + (ptrdiff_t)cu_y * j->ref_stride + cu_x, j->ref_stride, &p,
                   j->out_x + (size_t)i * HMO_NUM_CTU_PARTS, j->out_y + (size_t)i * HMO_NUM_CTU_PARTS,
                   j->out_sad + (size_t)i * HMO_NUM_CTU_PARTS, NULL);
  }
  return NULL;
}

int hmo_search_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride,
                     int pic_w, int pic_h, int sr, const int16_t* pred_q, uint32_t lambda_q16,
                     int fen, int bit_depth, int ctu_first, int ctu_count, int n_threads,
                     int32_t* out_x, int32_t* out_y, uint32_t* out_sad) {
  const int ctus_x = (pic_w + HMO_CTU - 1) / HMO_CTU, ctus_y = (pic_h + HMO_CTU - 1) / HMO_CTU;
  if (ctu_count < 0) ctu_count = ctus_x * ctus_y - ctu_first;
  if (n_threads <= 0) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  frame_job jobs[256];
  pthread_t th[256];
  for (int t = 0; t < n_threads; ++t) {
    frame_job j = {cur, cur_stride, ref, ref_stride, pic_w, pic_h, sr, pred_q, lambda_q16, fen, bit_depth,
                   ctu_first, ctu_count, t, n_threads, out_x, out_y, out_sad};
    jobs[t] = j;
  }
  if (n_threads == 1) { frame_worker(&jobs[0]); return ctu_count; }
  for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, frame_worker, &jobs[t]);
  for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  return ctu_count;
}

/* ------------------------------------------------------------------------------------ */
/* CPU baseline: HM's default fast search (xTZSearch) over every PU shape of every CTU    */
/* ------------------------------------------------------------------------------------ */
typedef struct tz_job {
  const hmo_pel* cur; int cur_stride; const hmo_pel* ref; int ref_stride;
  int pic_w, pic_h, sr; const int16_t* pred_q; uint32_t lambda_q16; int fen, bit_depth;
  int ctu_first, ctu_count, tid, n_threads, all_slots;
  int32_t *out_x, *out_y; uint32_t* out_sad;
  long probes; double sad4x4;
} tz_job;

static void* tz_worker(void* arg) {
  tz_job* j = (tz_job*)arg;
  init_slots();
  const int ctus_x = (j->pic_w + HMO_CTU - 1) / HMO_CTU;
  for (int i = j->tid; i < j->ctu_count; i += j->n_threads) {
    const int ctu = j->ctu_first + i;
    const int cu_x = (ctu % ctus_x) * HMO_CTU, cu_y = (ctu / ctus_x) * HMO_CTU;
    hmo_params p;
    p.pred_x = j->pred_q ? j->pred_q[2 * ctu] : 0;
    p.pred_y = j->pred_q ? j->pred_q[2 * ctu + 1] : 0;
    p.lambda_q16 = j->lambda_q16; p.fen = j->fen; p.bit_depth = j->bit_depth;
    hmo_set_search_range(p.pred_x, p.pred_y, j->sr, cu_x, cu_y, j->pic_w, j->pic_h, HMO_CTU,
                         &p.lt_x, &p.lt_y, &p.rb_x, &p.rb_y);
    hmo_tz_ctx tz = {j->sr, cu_x, cu_y, j->pic_w, j->pic_h, HMO_CTU};
    int imv[2] = {0, 0};
    /* slot 592 (64x64 2Nx2N, no integer-MV predictor) first, then the other shapes seeded with its
     * result like m_integerMv2Nx2N (TEncSearch.cpp:3780-3789) */
    for (int n = 0; n < (j->all_slots ? HMO_NUM_CTU_PARTS : 1); ++n) {
      const int s = n == 0 ? 592 : n - 1;
      const hmo_rect r = g_slot_rect[s];
      const ptrdiff_t co = (ptrdiff_t)(cu_y + r.y) * j->cur_stride + cu_x + r.x;
      const ptrdiff_t ro = (ptrdiff_t)(cu_y + r.y) * j->ref_stride + cu_x + r.x;
      int mx, my; uint32_t sad;
      const long pr = hmo_tz_search(j->cur + co, j->cur_stride, r.w, r.h, j->ref + ro, j->ref_stride, &p, &tz,
                                    n == 0 ? NULL : imv, p.pred_x, p.pred_y, &mx, &my, &sad);
      if (n == 0) { imv[0] = mx; imv[1] = my; }
      j->probes += pr;
      j->sad4x4 += (double)pr * (r.w * r.h / 16.0) * ((j->fen && r.h > 8) ? 0.5 : 1.0);
      if (j->out_x) {
        j->out_x[(size_t)i * HMO_NUM_CTU_PARTS + s] = mx;
        j->out_y[(size_t)i * HMO_NUM_CTU_PARTS + s] = my;
        j->out_sad[(size_t)i * HMO_NUM_CTU_PARTS + s] = sad;
      }
    }
  }
  return NULL;
}

int hmo_tz_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride, int pic_w, int pic_h,
                 int sr, const int16_t* pred_q, uint32_t lambda_q16, int fen, int bit_depth, int ctu_first,
                 int ctu_count, int n_threads, int all_slots, int32_t* out_x, int32_t* out_y, uint32_t* out_sad,
                 long* probes, double* sad4x4) {
  const int ctus_x = (pic_w + HMO_CTU - 1) / HMO_CTU, ctus_y = (pic_h + HMO_CTU - 1) / HMO_CTU;
  if (ctu_count < 0) ctu_count = ctus_x * ctus_y - ctu_first;
  if (n_threads <= 0) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  static tz_job jobs[256];
  pthread_t th[256];
  for (int t = 0; t < n_threads; ++t) {
    tz_job j = {cur, cur_stride, ref, ref_stride, pic_w, pic_h, sr, pred_q, lambda_q16, fen, bit_depth,
                ctu_first, ctu_count, t, n_threads, all_slots, out_x, out_y, out_sad, 0, 0.0};
    jobs[t] = j;
  }
  if (n_threads == 1) tz_worker(&jobs[0]);
  else {
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, tz_worker, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  *probes = 0; *sad4x4 = 0.0;
  for (int t = 0; t < n_threads; ++t) { *probes += jobs[t].probes; *sad4x4 += jobs[t].sad4x4; }
  return ctu_count;
}

/* ------------------------------------------------------------------------------------ */
/* fractional-pel refinement: the step after the integer search                          */
/*   TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331), xPatternRefinement      */
/*   (:816-875, tables :51-75), xExtDIFUpSamplingH/Q (:5386-5600) over                     */
/*   TComInterpolationFilter (TComInterpolationFilter.cpp:57-63, :170-260) and             */
/*   TComRdCost::xGetHADs / xCalcHADs4x4 / xCalcHADs8x8 (TComRdCost.cpp:1343-1604)         */
/* ------------------------------------------------------------------------------------ */

/* luma filter taps, TComInterpolationFilter.cpp:57-63 */
static const int k_luma_filter[4][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};

/* Predicted w x h luma block at the quarter-pel displacement (qx, qy) from `ref` (the PU origin in the
 * reference plane).  What the search's m_filteredBlock tables hold: horizontal pass into 14-bit
 * intermediates (filter<8,false,true,false>), vertical pass with final rounding and clip
 * (filter<8,true,false,true>); fraction 0 = filterCopy, which is the same formula with taps {64}. */
void hmo_pred_block_qpel(const hmo_pel* ref, int ref_stride, int w, int h, int qx, int qy, int bit_depth,
                         hmo_pel* dst, int dst_stride) {
  const int ix = qx >> 2, fx = qx & 3, iy = qy >> 2, fy = qy & 3;
  const int head = (14 - bit_depth) > 2 ? (14 - bit_depth) : 2;   /* headRoom */
  const int sh1 = 6 - head;
  const int off1 = -(8192 << sh1);
  const int sh2 = 6 + head;
  const int off2 = (1 << (sh2 - 1)) + (8192 << 6);
  const int maxv = (1 << bit_depth) - 1;
  const int* ch = k_luma_filter[fx];
  const int* cv = k_luma_filter[fy];
  static __thread hmo_pel tmp[(64 + 7) * 64];
  const hmo_pel* src = ref + (ptrdiff_t)(iy - 3) * ref_stride + ix;
  for (int r = 0; r < h + 7; ++r) {
    for (int c = 0; c < w; ++c) {
      int sum = 0;
      for (int k = 0; k < 8; ++k) sum += ch[k] * src[c + k - 3];
      tmp[r * 64 + c] = (hmo_pel)((sum + off1) >> sh1);
    }
    src += ref_stride;
  }
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int sum = 0;
      for (int k = 0; k < 8; ++k) sum += cv[k] * tmp[(r + k) * 64 + c];
      int v = (sum + off2) >> sh2;
      v = v < 0 ? 0 : (v > maxv ? maxv : v);
      dst[r * dst_stride + c] = (hmo_pel)v;
    }
}

/* TComRdCost::xCalcHADs4x4 (TComRdCost.cpp:1343-1437): sum |4x4 Hadamard(org - cur)|, (satd + 1) >> 1 */
static uint32_t had4x4(const hmo_pel* o, int os, const hmo_pel* c, int cs) {
  int d[16], m[16];
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 4; ++k) d[4 * r + k] = o[r * os + k] - c[r * cs + k];
  for (int r = 0; r < 4; ++r) {   /* rows */
    const int a = d[4 * r] + d[4 * r + 3], b = d[4 * r + 1] + d[4 * r + 2], e = d[4 * r + 1] - d[4 * r + 2], f = d[4 * r] - d[4 * r + 3];
    m[4 * r] = a + b; m[4 * r + 1] = a - b; m[4 * r + 2] = f + e; m[4 * r + 3] = f - e;
  }
  uint32_t satd = 0;
  for (int k = 0; k < 4; ++k) {   /* columns */
    const int a = m[k] + m[12 + k], b = m[4 + k] + m[8 + k], e = m[4 + k] - m[8 + k], f = m[k] - m[12 + k];
    satd += (uint32_t)(abs(a + b) + abs(a - b) + abs(f + e) + abs(f - e));
  }
  return (satd + 1) >> 1;
}

/* TComRdCost::xCalcHADs8x8 (TComRdCost.cpp:1439-1534): sum |8x8 Hadamard(org - cur)|, (sad + 2) >> 2 */
static uint32_t had8x8(const hmo_pel* o, int os, const hmo_pel* c, int cs) {
  int m[64];
  for (int r = 0; r < 8; ++r)
    for (int k = 0; k < 8; ++k) m[8 * r + k] = o[r * os + k] - c[r * cs + k];
  for (int pass = 0; pass < 2; ++pass) {          /* unnormalised Walsh-Hadamard along rows, then columns */
    const int stride_e = pass ? 8 : 1, stride_v = pass ? 1 : 8;
    for (int v = 0; v < 8; ++v)
      for (int len = 1; len < 8; len <<= 1)
        for (int i = 0; i < 8; i += 2 * len)
          for (int j = i; j < i + len; ++j) {
            int* a = &m[v * stride_v + j * stride_e];
            int* b = &m[v * stride_v + (j + len) * stride_e];
            const int x = *a, y = *b;
            *a = x + y; *b = x - y;
          }
  }
  uint32_t sad = 0;
  for (int i = 0; i < 64; ++i) sad += (uint32_t)abs(m[i]);
  return (sad + 2) >> 2;
}

/* TComRdCost::xGetHADs (TComRdCost.cpp:1537-1604) for iStep == 1 */
uint32_t hmo_had(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride, int w, int h, int bit_depth) {
  uint32_t sum = 0;
  if ((h % 8 == 0) && (w % 8 == 0)) {
    for (int y = 0; y < h; y += 8)
      for (int x = 0; x < w; x += 8) sum += had8x8(org + y * org_stride + x, org_stride, cur + y * cur_stride + x, cur_stride);
  } else {
    for (int y = 0; y < h; y += 4)
      for (int x = 0; x < w; x += 4) sum += had4x4(org + y * org_stride + x, org_stride, cur + y * cur_stride + x, cur_stride);
  }
  return sum >> (bit_depth - 8);
}

/* refinement point orders, TEncSearch.cpp:51-75 */
static const int k_refine_h[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
static const int k_refine_q[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}};

/* xPatternSearchFracDIF for one PU.  `ref` = reference plane at the PU origin; (int_x, int_y) = the integer MV.
 * out: half-pel offset (-1..1), quarter-pel offset (-1..1), cost of the winner (distortion + MV cost at scale 0).
 * The final quarter-pel MV is (int << 2) + (half << 1) + qter (TEncSearch.cpp:3800-3803). */
/* wp != NULL: the slice carries explicit weighted prediction -- the distortion functions of xPatternRefinement are then xGetHADsw /
 * xGetSADw (TComRdCostWeightPrediction.cpp:407-470, :55-90): the interpolated (and clipped) prediction is weighted sample by sample,
 * `Pel pred = ((w0 * cur + round) >> shift) + offset`, before the difference is taken; everything else is unchanged */
static void weight_block(hmo_pel* blk, int stride, int w, int h, const hmo_wp* wp) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) blk[y * stride + x] = (hmo_pel)wp_pred(blk[y * stride + x], wp);
}
static void frac_refine_impl(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, int int_x,
                             int int_y, int pred_x, int pred_y, uint32_t lambda_q16, int use_had, int bit_depth, const hmo_wp* wp, int* half_x,
                             int* half_y, int* qter_x, int* qter_y, uint32_t* cost) {
  hmo_pel blk[64 * 64];
  uint32_t best = UINT_MAX;
  int bi = 0;
  /* half-pel stage: xPatternRefinement(iFrac = 2), MV cost at scale 1 around rcMvHalf = int << 1 */
  for (int i = 0; i < 9; ++i) {
    const int hx = k_refine_h[i][0], hy = k_refine_h[i][1];
    hmo_pred_block_qpel(ref, ref_stride, w, h, 4 * int_x + 2 * hx, 4 * int_y + 2 * hy, bit_depth, blk, 64);
    if (wp) weight_block(blk, 64, w, h, wp);
    uint32_t d = use_had ? hmo_had(org, org_stride, blk, 64, w, h, bit_depth) : hmo_sad(org, org_stride, blk, 64, w, h, 0, bit_depth);
    d += hmo_mv_cost(lambda_q16, 2 * int_x + hx, 2 * int_y + hy, pred_x, pred_y, 1);
    if (d < best) { best = d; bi = i; }
  }
  *half_x = k_refine_h[bi][0]; *half_y = k_refine_h[bi][1];
  /* quarter-pel stage: xPatternRefinement(iFrac = 1), cost scale 0 around rcMvQter = ((int << 1) + half) << 1 */
  const int bx = 4 * int_x + 2 * *half_x, by = 4 * int_y + 2 * *half_y;
  best = UINT_MAX; bi = 0;
  for (int i = 0; i < 9; ++i) {
    const int qx = k_refine_q[i][0], qy = k_refine_q[i][1];
    hmo_pred_block_qpel(ref, ref_stride, w, h, bx + qx, by + qy, bit_depth, blk, 64);
    if (wp) weight_block(blk, 64, w, h, wp);
    uint32_t d = use_had ? hmo_had(org, org_stride, blk, 64, w, h, bit_depth) : hmo_sad(org, org_stride, blk, 64, w, h, 0, bit_depth);
    d += hmo_mv_cost(lambda_q16, bx + qx, by + qy, pred_x, pred_y, 0);
    if (d < best) { best = d; bi = i; }
  }
  *qter_x = k_refine_q[bi][0]; *qter_y = k_refine_q[bi][1];
  *cost = best;
}

void hmo_frac_refine(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, int int_x,
                     int int_y, int pred_x, int pred_y, uint32_t lambda_q16, int use_had, int bit_depth, int* half_x,
                     int* half_y, int* qter_x, int* qter_y, uint32_t* cost) {
  frac_refine_impl(org, org_stride, w, h, ref, ref_stride, int_x, int_y, pred_x, pred_y, lambda_q16, use_had, bit_depth, NULL, half_x, half_y, qter_x,
                   qter_y, cost);
}
void hmo_frac_refine_w(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, int int_x, int int_y, int pred_x,
                       int pred_y, uint32_t lambda_q16, int use_had, int bit_depth, const hmo_wp* wp, int* half_x, int* half_y, int* qter_x,
                       int* qter_y, uint32_t* cost) {
  frac_refine_impl(org, org_stride, w, h, ref, ref_stride, int_x, int_y, pred_x, pred_y, lambda_q16, use_had, bit_depth, wp, half_x, half_y, qter_x,
                   qter_y, cost);
}

/* ---- whole-picture refinement: hmo_frac_refine for every slot of every CTU in [ctu_first, ctu_first + ctu_count), threaded over
 *      CTUs.  int_mv: [ctu_count][593][2] integer MVs (as hmo_search_frame / the engine produce them); pred_q: [n_ctu][2] by CTU
 *      raster address or NULL.  out_qmv: [ctu_count][593][2] quarter-pel MVs (int << 2) + (half << 1) + quarter
 *      (TEncSearch.cpp:3800-3803); out_cost: [ctu_count][593]. */
typedef struct {
  const hmo_pel *cur, *ref;
  int cur_stride, ref_stride, pic_w, pic_h;
  const int16_t* pred_q;
  uint32_t lambda_q16;
  int use_had, bit_depth, ctu_first, ctu_count, tid, n_threads;
  const int16_t* int_mv;
  int16_t* out_qmv;
  uint32_t* out_cost;
} refine_job;

static void* refine_worker(void* arg) {
  const refine_job* j = (const refine_job*)arg;
  const int ctus_x = (j->pic_w + HMO_CTU - 1) / HMO_CTU;
  for (int i = j->tid; i < j->ctu_count; i += j->n_threads) {
    const int ctu = j->ctu_first + i, cx = (ctu % ctus_x) * HMO_CTU, cy = (ctu / ctus_x) * HMO_CTU;
    const int px = j->pred_q ? j->pred_q[2 * ctu] : 0, py = j->pred_q ? j->pred_q[2 * ctu + 1] : 0;
    for (int s = 0; s < HMO_NUM_CTU_PARTS; ++s) {
      hmo_rect r;
      hmo_slot_rect(s, &r);
      const long o = (long)i * HMO_NUM_CTU_PARTS + s;
      const int ix = j->int_mv[2 * o], iy = j->int_mv[2 * o + 1];
      int hx, hy, qx, qy;
      uint32_t cost;
      hmo_frac_refine(j->cur + (long)(cy + r.y) * j->cur_stride + cx + r.x, j->cur_stride, r.w, r.h,
                      j->ref + (long)(cy + r.y) * j->ref_stride + cx + r.x, j->ref_stride, ix, iy, px, py, j->lambda_q16, j->use_had,
                      j->bit_depth, &hx, &hy, &qx, &qy, &cost);
      j->out_qmv[2 * o] = (int16_t)(4 * ix + 2 * hx + qx);
      j->out_qmv[2 * o + 1] = (int16_t)(4 * iy + 2 * hy + qy);
      j->out_cost[o] = cost;
    }
  }
  return NULL;
}

int hmo_refine_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride, int pic_w, int pic_h, const int16_t* pred_q,
                     uint32_t lambda_q16, int use_had, int bit_depth, int ctu_first, int ctu_count, int n_threads, const int16_t* int_mv,
                     int16_t* out_qmv, uint32_t* out_cost) {
  const int ctus_x = (pic_w + HMO_CTU - 1) / HMO_CTU, ctus_y = (pic_h + HMO_CTU - 1) / HMO_CTU;
  if (ctu_count < 0) ctu_count = ctus_x * ctus_y - ctu_first;
  if (n_threads <= 0) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  refine_job jobs[256];
  pthread_t th[256];
  for (int t = 0; t < n_threads; ++t) {
    refine_job j = {cur, ref, cur_stride, ref_stride, pic_w, pic_h, pred_q, lambda_q16, use_had, bit_depth, ctu_first, ctu_count, t, n_threads,
                    int_mv, out_qmv, out_cost};
    jobs[t] = j;
  }
  if (n_threads == 1) { refine_worker(&jobs[0]); return ctu_count; }
  for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, refine_worker, &jobs[t]);
  for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  return ctu_count;
}
