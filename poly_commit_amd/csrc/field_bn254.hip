// Everything templated on the scalar field, instantiated for Fr of bn254 (see pc_internal.hpp).
#include "field_ops_impl.hpp"
namespace pc {
const FieldOps& field_ops_bn254() { static const FieldOps t = FieldOpsImpl<pc_bn254_fr>::table(); return t; }
}
