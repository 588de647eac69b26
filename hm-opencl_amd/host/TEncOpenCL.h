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
  Distortion* getRuiCost() { return m_tab[m_bi].cost; }
  Int* getX() { return m_tab[m_bi].x; }
  Int* getY() { return m_tab[m_bi].y; }
  Void setLambda(Double lambda);
  Void setEnabled(Bool e) { m_enabled = e; }

  // ---- additive: CPU-search arithmetic ----
  Void setCostMode(CostMode m) { m_mode = m; }
  Void setPredictor(const TComMv& pred) { m_pred = pred; }          // m_pcRdCost->setPredictor, TEncSearch.cpp:3737
  Void setSearchRangeRB(const TComMv& rb) { m_rb = rb; }            // cMvSrchRngRB, TEncSearch.cpp:3732
  Void setFastEnc(Bool b) { m_fen = b; }                            // getUseFastEnc(), TEncSearch.cpp:3853
  /// sample bit depth (SPS).  0 = unknown (the default: nothing in the reference tree tells this class): ME_MODE_OCL_COMPAT
  /// then derives the sample width from the samples of each call (reference window and current block), keeps the largest seen so
  /// far and, like cl/sad.cl, never shifts the SAD; ME_MODE_HM assumes 8.
  Void setBitDepth(Int b) { m_bitDepth = b; }
  Int getInferredBitDepth() const { return m_inferredDepth; }
  /// results in TComMv layout, ready for memcpy into TEncSearch::allMotionVectors[list][refIdx]
  const TComMv* getMvs() const { return m_tab[m_bi].mv; }
  Bool lastCallOk() const { return m_lastOk; }

  // ---- additive: explicit weighted prediction ----
  /// A slice with weighted prediction searches with m_cDistParam.bApplyWeight (setWpScalingDistParam, TEncSearch.cpp:3740, :5594-5635):
  /// every candidate is priced by xGetSADw on ((w * ref + round) >> shift) + offset.  While a weight is set, ME_MODE_HM calls run
  /// hmme_search_ctu_w -- with setRefine on, hmme_search_refine_ctu_w: the refinement prices xGetHADsw / xGetSADw like HM's own; a call
  /// the engine cannot serve exactly (weighted samples beyond a Pel, sums beyond its cost field) fails like any other -- lastCallOk()
  /// false, tables poisoned -- and the caller searches on the CPU.
  Void setWeight(Int w, Int offset, Int shift, Int round) { m_wpOn = true; m_wp[0] = w; m_wp[1] = offset; m_wp[2] = shift; m_wp[3] = round; }
  Void clearWeight() { m_wpOn = false; }
  Bool getWeightOn() const { return m_wpOn; }

  // ---- additive: bi-prediction refinement ----
  /// The bi-prediction pass (bBi, TEncSearch.cpp:3221) searches 2*org - pred_other with BipredSearchRange around the
  /// uni-prediction winner.  While setBiPred(true), calcMotionVectors fills -- and the getters return -- a SECOND table set,
  /// so the uni-prediction tables of that [list][refIdx] (filled at TEncSearch.cpp:3760-3764) are not clobbered.
  Void setBiPred(Bool b) { m_bi = b ? 1 : 0; }
  Bool getBiPred() const { return m_bi != 0; }
  const TComMv* getMvs(Bool bi) const { return m_tab[bi ? 1 : 0].mv; }
  const Distortion* getRuiCost(Bool bi) const { return m_tab[bi ? 1 : 0].cost; }

  // ---- additive: fractional-pel refinement tables ----
  /// While on, every ME_MODE_HM call also runs TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331) for the
  /// 593 slots on the device (one engine call, hmme_search_refine_ctu): getQMvs() / getFracCost() then hold, per slot, the
  /// quarter-pel MV (int << 2) + (half << 1) + quarter and the ruiCost that function returns, MV cost priced against the CTU's
  /// predictor.  fracOk() tells whether the last call produced them (failed calls do not).  A bi-prediction call (setBiPred(true):
  /// the block is 2*org - pred_other) is refined as well -- HM's bBi pass runs xPatternSearchFracDIF on that origin
  /// (TEncSearch.cpp:3798) -- and fracWasBi() tells which kind the tables of the last call belong to.
  Void setRefine(Bool on, Bool hadamard = true) { m_refine = on; m_refineHad = hadamard; }
  Bool fracOk() const { return m_fracOk; }
  Bool fracWasBi() const { return m_fracBi; }
  const TComMv* getQMvs() const { return m_qmv; }
  const Distortion* getFracCost() const { return m_fracCost; }
  /// keep the refinement tables of the last call for [list][refIdx] (the caller's allMotionVectors / allRuiCost have no room for
  /// them, TEncSearch.h:114-115); the stored distortion is the cost minus the MV cost against the CTU predictor, so that a PU can
  /// add the MV cost against ITS predictor (TEncSearch.cpp:3806-3808 does the same with the bits)
  Void storeFrac(Int list, Int refIdx);
  Bool fracStored(Int list, Int refIdx, Int poc, Int ctuAddr) const {
    return tablesValidFor(list, refIdx, poc, ctuAddr) && m_fracTag[list][refIdx];
  }
  /// stored tables of [list][refIdx]; where none are stored (fracStored() false: the refinement failed, was never requested, or
  /// the indices are out of range) these return what xPoison writes -- MV (0,0) and the largest Distortion -- never stale or
  /// unallocated memory: a host must stay up on exactly the path where the engine reported an error
  const TComMv& getFracMv(Int list, Int refIdx, Int slot) const {
    return xFracReadable(list, refIdx, slot) ? m_fracMvTab[list][refIdx][slot] : m_zeroMv;
  }
  Distortion getFracDist(Int list, Int refIdx, Int slot) const {
    return xFracReadable(list, refIdx, slot) ? m_fracDistTab[list][refIdx][slot] : (Distortion)~(Distortion)0;
  }
  Distortion getFracCostStored(Int list, Int refIdx, Int slot) const {
    return xFracReadable(list, refIdx, slot) ? m_fracCostTab[list][refIdx][slot] : (Distortion)~(Distortion)0;
  }
  /// HMME_GPU_FRAC=1: the patched encoder serves xPatternSearchFracDIF (TEncSearch.cpp:3798) from these tables
  static Bool gpuFracEnabled();

  // ---- additive: picture-edge CTUs ----
  /// A CTU that crosses the picture border is never coded as one 64x64 CU (TEncCu.cpp:424-425), so the reference never
  /// sends it to the GPU and its sub-CUs look up the PREVIOUS CTU's tables.  This runs the same search for such a CTU:
  /// pelCtuInPic = original picture at the CTU origin (only validW x validH samples are read; the rest of the 64x64
  /// block replicates the last valid column / row), window from xSetSearchRange + clipMv at the CTU origin
  /// (ME_MODE_HM arithmetic, predictor = pred).  Only slots inside the picture are meaningful -- the only ones HM looks up.
  Void calcMotionVectorsEdge(const Pel* pelCtuInPic, Int iPicStride, Int validW, Int validH, Pel* pelSearch, Int iRefStride,
                             Int i_areaSize, const TComMv& pred, Int ctuX, Int ctuY, Int picW, Int picH);
  /// which (picture, CTU) the caller's tables of [list][refIdx] were computed for (the caller owns the tables,
  /// TEncSearch.h:114-115; the tags live here so that TEncSearch.h stays untouched)
  Bool tablesValidFor(Int list, Int refIdx, Int poc, Int ctuAddr) const {
    return list >= 0 && list < 2 && refIdx >= 0 && refIdx < 33 && m_tagPoc[list][refIdx] == poc && m_tagCtu[list][refIdx] == ctuAddr;
  }
  Void markTables(Int list, Int refIdx, Int poc, Int ctuAddr) {
    if (list >= 0 && list < 2 && refIdx >= 0 && refIdx < 33) { m_tagPoc[list][refIdx] = poc; m_tagCtu[list][refIdx] = ctuAddr; m_fracTag[list][refIdx] = false; }
  }
  long numCalls() const { return m_calls; }
  long numFailed() const { return m_failed; }
  double engineSeconds() const { return m_engineSeconds; }
  /// window of the last call (calcMotionVectorsEdge derives it itself)
  const TComMv& getLastLT() const { return m_lastLT; }
  const TComMv& getLastRB() const { return m_lastRB; }
  /// A/B and self-check switches of the in-tree integration (tools/hm_patch): HMME_HM_MODE=0 keeps the reference call
  /// sequence (ME_MODE_OCL_COMPAT) in a patched encoder; HMME_VERIFY=1 makes the patched encoder re-run HM's own CPU
  /// xPatternSearch beside engine results and report the comparison through noteVerify (summary line in the destructor)
  /// rectangle of table slot 0..592 inside the CTU (TComDataCU::getIndexBlock order); false if out of range
  static Bool slotRect(Int slot, Int& x, Int& y, Int& w, Int& h);
  static Bool hmModeEnabled();
  static Bool verifyEnabled();
  Void noteVerify(Bool match) { ++m_verified; if (!match) ++m_verifyFailed; }
  long numVerified() const { return m_verified; }
  long numVerifyFailed() const { return m_verifyFailed; }

 protected:
  struct Tables {
    Int x[NUM_CTU_PARTS], y[NUM_CTU_PARTS];
    Distortion cost[NUM_CTU_PARTS];
    TComMv mv[NUM_CTU_PARTS];
  };
  Void xPoison(Tables& t);
  Bool xFracReadable(Int list, Int refIdx, Int slot) const {
    return m_fracMvTab && list >= 0 && list < 2 && refIdx >= 0 && refIdx < 33 && slot >= 0 && slot < NUM_CTU_PARTS && m_fracTag[list][refIdx];
  }

  hmme_ctx* m_ctx;
  Int m_deviceId;
  Bool m_deviceFound, m_enabled, m_lastOk;
  Int m_searchRange;
  CostMode m_mode;
  TComMv m_pred, m_rb;
  Bool m_fen;
  Int m_bitDepth;
  Int m_inferredDepth;                 // ME_MODE_OCL_COMPAT without setBitDepth: widest sample width seen so far (>= 8)
  Int m_bi;
  Double m_lambda;
  long m_calls, m_failed, m_edgeCalls, m_biCalls, m_verified, m_verifyFailed;
  TComMv m_lastLT, m_lastRB;
  Tables m_tab[2];                     // [0] uni-prediction, [1] bi-prediction refinement
  Int m_tagPoc[2][33], m_tagCtu[2][33];
  Bool m_refine, m_refineHad, m_fracOk, m_fracBi;
  // what the engine returns: always the 593-slot layout.  The tables above are NUM_CTU_PARTS entries in the caller's layout: the same
  // 593, or the 425 of an encoder built with AMP_ENC_SPEEDUP (TypeDef.h:260-261), filled through hmme_amp_off_slot
  TComMv m_engMv[593], m_engQmv[593];
  Distortion m_engCost[593], m_engFracCost[593];
  TComMv m_qmv[NUM_CTU_PARTS];
  Distortion m_fracCost[NUM_CTU_PARTS];
  TComMv m_fracPred;                   // predictor the refinement of the last call priced its MVs against
  Bool m_fracTag[2][33];
  TComMv m_zeroMv;                     // what the stored-table getters hand out when nothing is stored
  TComMv (*m_fracMvTab)[33][NUM_CTU_PARTS];          // [2][33][593], allocated on first use (313 KB + 2 x 157 KB)
  Distortion (*m_fracDistTab)[33][NUM_CTU_PARTS];
  Distortion (*m_fracCostTab)[33][NUM_CTU_PARTS];
  Bool m_wpOn;                         // explicit weighted prediction for the calls that follow (ME_MODE_HM)
  Int m_wp[4];                         // w, offset, shift, round of the reference picture's luma WPScalingParam
  long m_wpCalls;
  double m_engineSeconds;              // wall time spent inside the engine calls (HMME_TRACE summary)
};

#endif
