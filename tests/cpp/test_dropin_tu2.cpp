// Second translation unit that includes the drop-in headers: the reference's headers cannot be
// included twice in one program (a non-inline namespace-scope variable, reference
// src/types.hpp:10); these can.
#include "hom_nand.hpp"

int second_translation_unit_value() { return static_cast<int>(sizeof(TFHE::p)) * 0 + TFHE::TLWE<>::N + 1; }
