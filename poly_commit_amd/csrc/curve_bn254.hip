// Everything templated on the curve, instantiated for bn254 (see pc_internal.hpp).
#include "curve_ops_impl.hpp"
namespace pc {
const CurveOps& curve_ops_bn254() { static const CurveOps t = CurveOpsImpl<pc_curve_bn254>::table(); return t; }
}
