#pragma once
#include <memory>
namespace boost { template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A> shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(static_cast<A &&>(a)...); } }
