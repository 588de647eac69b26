// TEncOpenCL.cpp -- see TEncOpenCL.h.  Everything device-side happens behind include/hmme.h.
#include "TEncOpenCL.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/hmme.h"

TEncOpenCL::TEncOpenCL()
    : m_ctx(0), m_deviceId(0), m_deviceFound(false), m_enabled(false), m_lastOk(false), m_searchRange(0),
      m_mode(ME_MODE_OCL_COMPAT), m_fen(false), m_bitDepth(8), m_lambda(0.0), m_calls(0), m_failed(0) {
  std::memset(m_x, 0, sizeof m_x);
  std::memset(m_y, 0, sizeof m_y);
  std::memset(m_ruiCosts, 0, sizeof m_ruiCosts);
}

TEncOpenCL::~TEncOpenCL() {
  if (std::getenv("HMME_TRACE"))   // one summary line for A/B harnesses (tests/test_hm_dropin.py)
    fprintf(stderr, "TEncOpenCL(hmme): %ld calcMotionVectors calls, %ld failed, device: %s\n", m_calls, m_failed,
            m_ctx ? hmme_device_info(m_ctx) : "none");
  if (m_ctx) hmme_destroy(m_ctx);
  m_ctx = 0;
}

// reference: scans OpenCL platforms for GPUs and remembers the device (TEncOpenCL.cpp:69-137).  Here the
// device is validated when the context is created; a missing gfx950 GPU makes createBuffers return false.
Bool TEncOpenCL::findDevice(Int device) {
  m_deviceId = device;
  m_deviceFound = device >= 0;
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

// reference TEncOpenCL.cpp:240-362.  pelSearch = reference plane at the CTU origin, i_areaSize = search range,
// *pcMvSrchRngLT = integer-pel top-left of the window.
Void TEncOpenCL::calcMotionVectors(Pel* pelCtu, Pel* pelSearch, Int iRefStride, Int iCtuStride, Int i_areaSize,
                                   TComMv* pcMvSrchRngLT) {
  m_lastOk = false;
  ++m_calls;
  ++m_failed;
  if (!m_ctx) {
    fprintf(stderr, "ERROR: TEncOpenCL::calcMotionVectors called without a device context\n");
    return;
  }
  hmme_search_params p;
  if (m_mode == ME_MODE_OCL_COMPAT) {
    hmme_params_ocl_compat(&p, pcMvSrchRngLT->getHor(), pcMvSrchRngLT->getVer(), i_areaSize);
  } else {
    p.lt_x = pcMvSrchRngLT->getHor(); p.lt_y = pcMvSrchRngLT->getVer();
    p.rb_x = m_rb.getHor(); p.rb_y = m_rb.getVer();
    p.pred_x = m_pred.getHor(); p.pred_y = m_pred.getVer();
    p.fen = m_fen ? 1 : 0;
  }
  p.bit_depth = m_bitDepth;
  typedef char tcommv_is_two_shorts[sizeof(TComMv) == 4 ? 1 : -1];   // C++98-friendly static assert
  (void)sizeof(tcommv_is_two_shorts);
  if (hmme_search_ctu(m_ctx, pelCtu, iCtuStride, pelSearch, iRefStride, &p, reinterpret_cast<int16_t*>(m_mv), m_ruiCosts) !=
      HMME_OK) {
    fprintf(stderr, "ERROR: TEncOpenCL::calcMotionVectors: %s\n", hmme_last_error(m_ctx));
    return;
  }
  for (Int i = 0; i < NUM_CTU_PARTS; i++) {   // the reference hands out Int arrays (TEncOpenCL.h:118-119)
    m_x[i] = m_mv[i].getHor();
    m_y[i] = m_mv[i].getVer();
  }
  m_lastOk = true;
  --m_failed;
}
