// Everything templated on the scalar field, instantiated for Fr of bls12_381 (see pc_internal.hpp).
#include "field_ops_impl.hpp"
namespace pc {
const FieldOps& field_ops_bls12_381() { static const FieldOps t = FieldOpsImpl<pc_bls12_381_fr>::table(); return t; }
}
