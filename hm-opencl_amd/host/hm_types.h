// hm_types.h -- the handful of HM 16.4 types the TEncOpenCL surface uses, for building the host
// module OUTSIDE an HM tree (tests, tools).  Inside HM, compile with -DHMME_IN_HM_TREE and the real
// TLibCommon/TypeDef.h + TComMv.h are used instead; the layouts below mirror them
// (reference source/Lib/TLibCommon/TypeDef.h:660-720, 263; TComMv.h:51-150).
#ifndef HMME_HM_TYPES_H
#define HMME_HM_TYPES_H

typedef void Void;
typedef bool Bool;
typedef char Char;
typedef int Int;
typedef unsigned int UInt;
typedef short Short;
typedef double Double;
typedef Short Pel;            // TypeDef.h:706 (RExt__HIGH_BIT_DEPTH_SUPPORT == 0)
typedef UInt Distortion;      // TypeDef.h:717 (FULL_NBIT == 0)
#ifndef NUM_CTU_PARTS
#define NUM_CTU_PARTS 593     // TypeDef.h:263 (425 in a build with AMP_ENC_SPEEDUP, TypeDef.h:260-261: -DNUM_CTU_PARTS=425)
#endif

class TComMv {                // {Short hor, Short ver}, TComMv.h:51-55
 public:
  TComMv() : m_iHor(0), m_iVer(0) {}
  TComMv(Short h, Short v) : m_iHor(h), m_iVer(v) {}
  Void set(Short h, Short v) { m_iHor = h; m_iVer = v; }
  Void setHor(Short h) { m_iHor = h; }
  Void setVer(Short v) { m_iVer = v; }
  Int getHor() const { return m_iHor; }
  Int getVer() const { return m_iVer; }
 private:
  Short m_iHor, m_iVer;
};

#endif
