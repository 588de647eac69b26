// TEncOpenCL.cpp -- see TEncOpenCL.h.  Everything device-side happens behind include/hmme.h.
#include "TEncOpenCL.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>

#include "../../include/hmme.h"

namespace {
double nowSeconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace

TEncOpenCL::TEncOpenCL()
    : m_ctx(0), m_deviceId(0), m_deviceFound(false), m_enabled(false), m_lastOk(false), m_searchRange(0),
      m_mode(ME_MODE_OCL_COMPAT), m_fen(false), m_bitDepth(0), m_inferredDepth(8), m_bi(0), m_lambda(0.0), m_calls(0), m_failed(0), m_edgeCalls(0),
      m_biCalls(0), m_verified(0), m_verifyFailed(0), m_refine(false), m_refineHad(true), m_fracOk(false), m_fracBi(false),
      m_fracMvTab(0), m_fracDistTab(0), m_fracCostTab(0), m_wpOn(false), m_wpCalls(0), m_engineSeconds(0.0) {
  m_wp[0] = 1; m_wp[1] = m_wp[2] = m_wp[3] = 0;
  for (Int b = 0; b < 2; b++) {
    xPoison(m_tab[b]);   // tables nobody filled yet must not look like results either
  }
  for (Int l = 0; l < 2; l++)
    for (Int r = 0; r < 33; r++) { m_tagPoc[l][r] = -0x7fffffff; m_tagCtu[l][r] = -1; m_fracTag[l][r] = false; }
}

TEncOpenCL::~TEncOpenCL() {
  if (std::getenv("HMME_TRACE"))   // one summary line for A/B harnesses (tests/test_hm_dropin.py)
    fprintf(stderr, "TEncOpenCL(hmme): %ld calcMotionVectors calls, %ld failed, %ld edge-CTU, %ld bi-pred, %ld results verified against xPatternSearch, "
            "%ld differ, %.3f s inside the engine calls (%ld weighted), device: %s\n", m_calls, m_failed, m_edgeCalls, m_biCalls, m_verified, m_verifyFailed,
            m_engineSeconds, m_wpCalls, m_ctx ? hmme_device_info(m_ctx) : "none");
  if (m_ctx) hmme_destroy(m_ctx);
  m_ctx = 0;
  delete[] m_fracMvTab; delete[] m_fracDistTab; delete[] m_fracCostTab;
}

Bool TEncOpenCL::slotRect(Int slot, Int& x, Int& y, Int& w, Int& h) {   // `slot` in the caller's layout
  return hmme_slot_rect(NUM_CTU_PARTS == 593 ? slot : hmme_amp_off_slot(slot), &x, &y, &w, &h) == HMME_OK;
}
Bool TEncOpenCL::hmModeEnabled() {
  static const Int on = (std::getenv("HMME_HM_MODE") && std::getenv("HMME_HM_MODE")[0] == '0') ? 0 : 1;
  return on != 0;
}
Bool TEncOpenCL::gpuFracEnabled() {
  static const Int on = (std::getenv("HMME_GPU_FRAC") && std::getenv("HMME_GPU_FRAC")[0] == '1') ? 1 : 0;
  return on != 0;
}
Bool TEncOpenCL::verifyEnabled() {
  static const Int on = (std::getenv("HMME_VERIFY") && std::getenv("HMME_VERIFY")[0] == '1') ? 1 : 0;
  return on != 0;
}

// reference: scans OpenCL platforms for GPUs and remembers the device (TEncOpenCL.cpp:69-137).  Here the
// device is validated when the context is created; a missing gfx950 GPU makes createBuffers return false.
Bool TEncOpenCL::findDevice(Int device) {
  m_deviceId = device;
  m_deviceFound = device >= 0;
  if (hmme_abi_version() != HMME_ABI_VERSION) {   // a libhmme.so older or newer than the header this file was compiled against
    fprintf(stderr, "ERROR: TEncOpenCL::findDevice: libhmme ABI version %d, this module was built for %d\n", hmme_abi_version(), HMME_ABI_VERSION);
    m_deviceFound = false;
  }
  return m_deviceFound;
}

// reference: reads and JIT-compiles cl/sad.cl (TEncOpenCL.cpp:139-190).  The HIP code object is embedded
// in libhmme.so, so there is nothing to compile; the arguments are accepted for source compatibility.
Bool TEncOpenCL::compileKernelSource(const Char* /*fileName*/, const Char* /*kernelNameCalc*/) { return m_deviceFound; }

// reference: allocates CTU / window / result buffers for the given search range (TEncOpenCL.cpp:195-238)
Bool TEncOpenCL::createBuffers(UInt maxCtuWidth, UInt maxCtuHeight, Int searchRange) {
  if (maxCtuWidth != HMME_CTU_SIZE || maxCtuHeight != HMME_CTU_SIZE) {
    fprintf(stderr, "ERROR: TEncOpenCL::createBuffers: CTU %ux%u unsupported (64x64 only, like cl/sad.cl)\n", maxCtuWidth,
            maxCtuHeight);
    return false;
  }
  if (m_ctx) { hmme_destroy(m_ctx); m_ctx = 0; }
  m_searchRange = searchRange;
  if (hmme_create(m_deviceId, searchRange, 0, &m_ctx) != HMME_OK) {
    fprintf(stderr, "ERROR: TEncOpenCL::createBuffers: %s\n", hmme_last_error(0));
    m_ctx = 0;
    return false;
  }
  hmme_set_lambda(m_ctx, m_lambda);
  return true;
}

const Char* TEncOpenCL::getDeviceInfo() { return m_ctx ? hmme_device_info(m_ctx) : ""; }

Void TEncOpenCL::setLambda(Double lambda) {   // reference TEncOpenCL.h:121
  m_lambda = lambda;
  if (m_ctx) hmme_set_lambda(m_ctx, lambda);
}

// A failed call must never leave the previous CTU's results behind: the reference caller copies the tables without looking
// at any status (TEncSearch.cpp:3749-3765).  MV (0,0) and the largest Distortion cannot win a comparison by accident.
Void TEncOpenCL::xPoison(Tables& t) {
  for (Int i = 0; i < NUM_CTU_PARTS; i++) {
    t.x[i] = 0; t.y[i] = 0;
    t.cost[i] = (Distortion)~(Distortion)0;
    t.mv[i].set(0, 0);
  }
}

// row reductions of the compat-mode sample-width scan, vectorised whatever the optimisation level of the build they end up in (the
// reference's own makefiles compile the encoder at -O2, where GCC 11 does not vectorise)
#if defined(__GNUC__) && !defined(__clang__)
#define HMME_VECTORISE __attribute__((optimize("O3", "tree-vectorize")))
#else
#define HMME_VECTORISE
#endif
static HMME_VECTORISE Pel xRowMax(const Pel* __restrict__ row, Int n, Pel m) {
  for (Int x = 0; x < n; x++) m = row[x] > m ? row[x] : m;
  return m;
}
static HMME_VECTORISE Pel xRowMin(const Pel* __restrict__ row, Int n, Pel m) {
  for (Int x = 0; x < n; x++) m = row[x] < m ? row[x] : m;
  return m;
}

// reference TEncOpenCL.cpp:240-362.  pelSearch = reference plane at the CTU origin, i_areaSize = search range,
// *pcMvSrchRngLT = integer-pel top-left of the window.
Void TEncOpenCL::calcMotionVectors(Pel* pelCtu, Pel* pelSearch, Int iRefStride, Int iCtuStride, Int i_areaSize,
                                   TComMv* pcMvSrchRngLT) {
  Tables& t = m_tab[m_bi];
  m_lastOk = false;
  ++m_calls;
  ++m_failed;
  if (m_bi) ++m_biCalls;
  if (!m_ctx) {
    fprintf(stderr, "ERROR: TEncOpenCL::calcMotionVectors called without a device context\n");
    xPoison(t);
    return;
  }
  m_lastLT = *pcMvSrchRngLT;
  hmme_search_params p;
  if (m_mode == ME_MODE_OCL_COMPAT) {
    hmme_params_ocl_compat(&p, pcMvSrchRngLT->getHor(), pcMvSrchRngLT->getVer(), i_areaSize);
    p.bit_depth = m_bitDepth;
    if (m_bitDepth <= 0) p.bit_depth = m_inferredDepth;   // the latched width; a call it is too narrow for comes back with HMME_ERR_RANGE (below)
  } else {
    p.lt_x = pcMvSrchRngLT->getHor(); p.lt_y = pcMvSrchRngLT->getVer();
    p.rb_x = m_rb.getHor(); p.rb_y = m_rb.getVer();
    p.pred_x = m_pred.getHor(); p.pred_y = m_pred.getVer();
    p.fen = m_fen ? 1 : 0;
    p.bit_depth = m_bitDepth > 0 ? m_bitDepth : 8;
    p.shift_free = 0;
  }
  m_lastRB.set((Short)p.rb_x, (Short)p.rb_y);
  typedef char tcommv_is_two_shorts[sizeof(TComMv) == 4 ? 1 : -1];   // C++98-friendly static asserts
  typedef char distortion_is_u32[sizeof(Distortion) == 4 ? 1 : -1];
  (void)sizeof(tcommv_is_two_shorts); (void)sizeof(distortion_is_u32);
  // refinement rides along on calls with HM's arithmetic, uni-prediction and bi-prediction origins alike (the bBi call of
  // TEncSearch.cpp:3798); where the engine cannot refine the integer search still runs and fracOk() says so
  const Bool weighted = m_wpOn && m_mode == ME_MODE_HM;   // the reference kernel knows no weights (cl/sad.cl): compat mode ignores them
  const Bool wantFrac = m_refine && m_mode == ME_MODE_HM;
  m_fracOk = false;
  m_fracBi = m_bi != 0;
  Int rc = HMME_ERR_UNSUPPORTED;
  const double t0 = nowSeconds();   // wall time of the engine calls of this object: what the offload costs the encoder (tools/hm_ab.py)
  struct Timer { double& acc; double t0; Timer(double& a, double t) : acc(a), t0(t) {} ~Timer() { acc += nowSeconds() - t0; } } timer(m_engineSeconds, t0);
  if (weighted) {   // no unweighted second try: a failed weighted call must reach the caller as failed
    ++m_wpCalls;
    hmme_weight w = {m_wp[0], m_wp[1], m_wp[2], m_wp[3]};
    if (wantFrac) {   // weighted search + weighted refinement (xGetHADsw) in one call; where the engine cannot refine, the search alone
      rc = hmme_search_refine_ctu_w(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, &w, m_refineHad ? 1 : 0, reinterpret_cast<int16_t*>(m_engMv),
                                    reinterpret_cast<uint32_t*>(m_engCost), reinterpret_cast<int16_t*>(m_engQmv), reinterpret_cast<uint32_t*>(m_engFracCost));
      m_fracOk = rc == HMME_OK;
      m_fracPred = m_pred;
    }
    if (rc != HMME_OK)
      rc = hmme_search_ctu_w(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, &w, reinterpret_cast<int16_t*>(m_engMv), reinterpret_cast<uint32_t*>(m_engCost));
    if (rc != HMME_OK) {
      fprintf(stderr, "ERROR: TEncOpenCL::calcMotionVectors (weighted prediction): %s\n", hmme_last_error(m_ctx));
      xPoison(t);
      return;
    }
  }
  if (wantFrac && !weighted) {
    rc = hmme_search_refine_ctu(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, m_refineHad ? 1 : 0, reinterpret_cast<int16_t*>(m_engMv),
                                reinterpret_cast<uint32_t*>(m_engCost), reinterpret_cast<int16_t*>(m_engQmv), reinterpret_cast<uint32_t*>(m_engFracCost));
    m_fracOk = rc == HMME_OK;
    m_fracPred = m_pred;
  }
  if (rc != HMME_OK && !weighted) {
    const Bool probing = m_mode == ME_MODE_OCL_COMPAT && m_bitDepth <= 0 && p.bit_depth < 12;
    const int printing = probing ? hmme_set_error_printing(m_ctx, 0) : 1;   // the caller's own setting comes back after the probe and its retry
    rc = hmme_search_ctu(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, reinterpret_cast<int16_t*>(m_engMv), reinterpret_cast<uint32_t*>(m_engCost));
    if (probing && rc != HMME_ERR_RANGE) hmme_set_error_printing(m_ctx, printing);
    if (probing && rc == HMME_ERR_RANGE) {
      // Nothing in the reference tree tells this class the bit depth (createBuffers has no such argument), and cl/sad.cl works on
      // whatever Pel holds without a shift: the sample width comes from the samples of the call -- the reference window AND the current
      // block (a dark window under a bright block, a fade or a cut, must not pick too narrow a width).  The block may be a bi-prediction
      // origin 2*org - pred (TEncSearch.cpp:3702-3712), so it only has to fit [-maxv, 2*maxv]: such a call never widens the estimate
      // beyond the true depth.  The width is latched (only ever grows), so one sequence does not alternate between the 8-bit and the
      // 16-bit kernel; with shift-free sums the results do not depend on it.  The engine looks at every sample of a call anyway and
      // refuses one that does not fit the width it was given: only such a call pays for the scan that finds the width it needs
      // (every call used to: 4 us beside a 55 us engine call, tools/class_latency.cpp).
      Pel hi16 = 0, chi16 = 0, clo16 = 0;
      const Int side = 2 * i_areaSize + 64;   // the window the reference copies, TEncOpenCL.cpp:253-277
      const Pel* row = pelSearch + (long)p.lt_y * iRefStride + p.lt_x;
      for (Int y = 0; y < side; y++, row += iRefStride) hi16 = xRowMax(row, side, hi16);
      row = pelCtu;
      for (Int y = 0; y < HMME_CTU_SIZE; y++, row += iCtuStride) { chi16 = xRowMax(row, HMME_CTU_SIZE, chi16); clo16 = xRowMin(row, HMME_CTU_SIZE, clo16); }
      const Int hi = hi16, chi = chi16, clo = clo16;
      Int d = m_inferredDepth;
      while (d < 12 && (hi > (1 << d) - 1 || chi > 2 * ((1 << d) - 1) || clo < -((1 << d) - 1))) ++d;
      m_inferredDepth = d;
      p.bit_depth = d;
      rc = hmme_search_ctu(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, reinterpret_cast<int16_t*>(m_engMv), reinterpret_cast<uint32_t*>(m_engCost));
      hmme_set_error_printing(m_ctx, printing);   // a failure of the retry is reported once, below, through hmme_last_error
    }
  }
  if (rc != HMME_OK) {
    fprintf(stderr, "ERROR: TEncOpenCL::calcMotionVectors: %s\n", hmme_last_error(m_ctx));
    xPoison(t);
    return;
  }
  typedef char known_table_size[(NUM_CTU_PARTS == 593 || NUM_CTU_PARTS == 425) ? 1 : -1];
  (void)sizeof(known_table_size);
  for (Int i = 0; i < NUM_CTU_PARTS; i++) {   // the caller's layout; the reference hands out Int arrays (TEncOpenCL.h:118-119)
    const Int s = NUM_CTU_PARTS == 593 ? i : hmme_amp_off_slot(i);
    t.mv[i] = m_engMv[s];
    t.cost[i] = m_engCost[s];
    t.x[i] = t.mv[i].getHor();
    t.y[i] = t.mv[i].getVer();
    if (m_fracOk) { m_qmv[i] = m_engQmv[s]; m_fracCost[i] = m_engFracCost[s]; }
  }
  m_lastOk = true;
  --m_failed;
}

Void TEncOpenCL::calcMotionVectorsEdge(const Pel* pelCtuInPic, Int iPicStride, Int validW, Int validH, Pel* pelSearch, Int iRefStride,
                                       Int i_areaSize, const TComMv& pred, Int ctuX, Int ctuY, Int picW, Int picH) {
  Pel block[HMME_CTU_SIZE * HMME_CTU_SIZE];
  validW = validW < 1 ? 1 : (validW > HMME_CTU_SIZE ? HMME_CTU_SIZE : validW);
  validH = validH < 1 ? 1 : (validH > HMME_CTU_SIZE ? HMME_CTU_SIZE : validH);
  for (Int y = 0; y < HMME_CTU_SIZE; y++) {
    const Pel* src = pelCtuInPic + (long)(y < validH ? y : validH - 1) * iPicStride;
    for (Int x = 0; x < HMME_CTU_SIZE; x++) block[y * HMME_CTU_SIZE + x] = src[x < validW ? x : validW - 1];
  }
  Int ltx, lty, rbx, rby;   // xSetSearchRange + clipMv at the CTU origin (TEncSearch.cpp:3814-3830, TComDataCU.cpp:2907-2920)
  hmme_set_search_range(pred.getHor(), pred.getVer(), i_areaSize, ctuX, ctuY, picW, picH, &ltx, &lty, &rbx, &rby);
  const CostMode mode = m_mode;
  const TComMv savePred = m_pred, saveRb = m_rb;
  TComMv lt((Short)ltx, (Short)lty);
  m_mode = ME_MODE_HM;
  m_pred = pred;
  m_rb.set((Short)rbx, (Short)rby);
  ++m_edgeCalls;
  calcMotionVectors(block, pelSearch, iRefStride, HMME_CTU_SIZE, i_areaSize, &lt);
  m_mode = mode; m_pred = savePred; m_rb = saveRb;
}

namespace {
// TComRdCost::xGetComponentBits (TComRdCost.cpp:278-292)
UInt componentBits(Int v) {
  UInt len = 1;
  UInt t = (v <= 0) ? ((UInt)(-v) << 1) + 1 : ((UInt)v << 1);
  while (t != 1) { t >>= 1; len += 2; }
  return len;
}
}  // namespace

Void TEncOpenCL::storeFrac(Int list, Int refIdx) {
  if (list < 0 || list > 1 || refIdx < 0 || refIdx >= 33) return;
  m_fracTag[list][refIdx] = false;
  if (!m_fracOk || !m_ctx) return;
  if (!m_fracMvTab) {
    m_fracMvTab = new TComMv[2][33][NUM_CTU_PARTS];
    m_fracDistTab = new Distortion[2][33][NUM_CTU_PARTS];
    m_fracCostTab = new Distortion[2][33][NUM_CTU_PARTS];
  }
  const UInt lq = hmme_get_lambda_q16(m_ctx);
  for (Int i = 0; i < NUM_CTU_PARTS; i++) {
    // TComRdCost::getCost(x, y) at cost scale 0 (TComRdCost.h:172-189): uint32 product, >> 16
    const UInt mvCost = (lq * (componentBits(m_qmv[i].getHor() - m_fracPred.getHor()) + componentBits(m_qmv[i].getVer() - m_fracPred.getVer()))) >> 16;
    m_fracMvTab[list][refIdx][i] = m_qmv[i];
    m_fracCostTab[list][refIdx][i] = m_fracCost[i];
    m_fracDistTab[list][refIdx][i] = m_fracCost[i] - mvCost;
  }
  m_fracTag[list][refIdx] = true;
}
