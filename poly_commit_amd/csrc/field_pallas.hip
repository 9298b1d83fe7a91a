// Everything templated on the scalar field, instantiated for Fr of pallas (see pc_internal.hpp).
#include "field_ops_impl.hpp"
namespace pc {
const FieldOps& field_ops_pallas() { static const FieldOps t = FieldOpsImpl<pc_pallas_fr>::table(); return t; }
}
