// Everything templated on the curve, instantiated for pallas (see pc_internal.hpp).
#include "curve_ops_impl.hpp"
namespace pc {
const CurveOps& curve_ops_pallas() { static const CurveOps t = CurveOpsImpl<pc_curve_pallas>::table(); return t; }
}
