// Everything templated on the curve, instantiated for bls12_381 (see pc_internal.hpp).
#include "curve_ops_impl.hpp"
namespace pc {
const CurveOps& curve_ops_bls12_381() { static const CurveOps t = CurveOpsImpl<pc_curve_bls12_381>::table(); return t; }
}
