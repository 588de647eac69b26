// TEncOpenCL.h -- MI355X host module with the public surface of the reference's OpenCL add-on
// (reference source/Lib/TLibEncoder/TEncOpenCL.h:105-127), implemented over the hmme C ABI
// (include/hmme.h) instead of OpenCL.  Dropping this header + TEncOpenCL.cpp into
// source/Lib/TLibEncoder and linking -lhmme -lamdhip64 instead of -lOpenCL leaves
// TEncTop::xInitOpenCL (TEncTop.cpp:1116-1162), TEncSlice (setLambda, TEncSlice.cpp:150) and
// TEncSearch::xMotionEstimation (TEncSearch.cpp:3743-3771) source-compatible.
//
// Same names, argument meaning and error behaviour as the reference:
//   * the three init calls return Bool; on false the caller disables the feature
//     (TEncTop.cpp:1131-1158) -- there is NO CPU fallback inside this module;
//   * calcMotionVectors is Void and reports errors on stderr (reference checkError, TEncOpenCL.h:93-101);
//   * getX()/getY()/getRuiCost() return module-owned arrays of NUM_CTU_PARTS entries in
//     TComDataCU::getIndexBlock order, valid until the next call.
// Additive (defaults reproduce the reference GPU path: predictor (0,0), window LT..LT+2*SR,
// all rows -- cl/sad.cl:374-408): setCostMode(ME_MODE_HM) + setPredictor / setSearchRangeRB /
// setFastEnc switch to the arithmetic of the CPU search TEncSearch::xPatternSearch.
#ifndef TENCOPENCL_H
#define TENCOPENCL_H

#ifdef HMME_IN_HM_TREE
#include "TLibCommon/TypeDef.h"
#include "TLibCommon/TComMv.h"
#else
#include "hm_types.h"
#endif

struct hmme_ctx;

class TEncOpenCL {
 public:
  enum CostMode { ME_MODE_OCL_COMPAT = 0, ME_MODE_HM = 1 };

  TEncOpenCL();
  virtual ~TEncOpenCL();
  Bool compileKernelSource(const Char* fileName, const Char* kernelNameCalc);
  Bool findDevice(Int device);
  Bool createBuffers(UInt i_maxCtuWidth, UInt i_maxCtuHeight, Int i_searchRange);
  Void calcMotionVectors(Pel* pelCtu, Pel* pelSearch, Int i_iRefStride, Int i_iCtuStride, Int i_areaSize,
                         TComMv* pcMvSrchRngLT);

  Int getDeviceId() { return m_deviceId; }
  Void setDeviceId(Int i) { m_deviceId = i; }
  const Char* getDeviceInfo();
  Distortion* getRuiCost() { return m_ruiCosts; }
  Int* getX() { return m_x; }
  Int* getY() { return m_y; }
  Void setLambda(Double lambda);
  Void setEnabled(Bool e) { m_enabled = e; }

  // ---- additive: CPU-search arithmetic ----
  Void setCostMode(CostMode m) { m_mode = m; }
  Void setPredictor(const TComMv& pred) { m_pred = pred; }          // m_pcRdCost->setPredictor, TEncSearch.cpp:3737
  Void setSearchRangeRB(const TComMv& rb) { m_rb = rb; }            // cMvSrchRngRB, TEncSearch.cpp:3732
  Void setFastEnc(Bool b) { m_fen = b; }                            // getUseFastEnc(), TEncSearch.cpp:3853
  Void setBitDepth(Int b) { m_bitDepth = b; }
  /// results in TComMv layout, ready for memcpy into TEncSearch::allMotionVectors[list][refIdx]
  const TComMv* getMvs() const { return m_mv; }
  Bool lastCallOk() const { return m_lastOk; }

 protected:
  hmme_ctx* m_ctx;
  Int m_deviceId;
  Bool m_deviceFound, m_enabled, m_lastOk;
  Int m_searchRange;
  CostMode m_mode;
  TComMv m_pred, m_rb;
  Bool m_fen;
  Int m_bitDepth;
  Double m_lambda;
  long m_calls, m_failed;
  Int m_x[NUM_CTU_PARTS], m_y[NUM_CTU_PARTS];
  Distortion m_ruiCosts[NUM_CTU_PARTS];
  TComMv m_mv[NUM_CTU_PARTS];
};

#endif
