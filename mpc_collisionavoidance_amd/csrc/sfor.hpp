// sfor.hpp — compile-time loop: every index is a constant expression, so register arrays stay in
// VGPRs (no scratch) and the DPP lane selectors of lanes.hpp can be template arguments.
#pragma once
#include <type_traits>

namespace usv {

template <int I, int E, class F>
USV_DEV void sfor(F &&f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}

} // namespace usv
