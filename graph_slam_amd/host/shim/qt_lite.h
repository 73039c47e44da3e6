// Stand-in for the handful of Qt / boost names the reference's wrappers use for the front-end thread pool
// (g2o/g2o_graph.cpp:14-18,196-226; gtsam/gtsam_graph.cpp:30-33,1717-1753): QList, QThreadPool::globalInstance,
// QtConcurrent::blockingMapped(list, boost::bind(&CCameraNode::matchNodePair, node, _1)).  The optimiser path does not
// depend on them; blockingMapped runs the calls in order on the caller's thread (the results are the same list).
#pragma once
#include <cstddef>
#include <vector>

template <class T>
class QList {
 public:
  void push_back(const T &v) { v_.push_back(v); }
  void append(const T &v) { v_.push_back(v); }
  int size() const { return (int)v_.size(); }
  bool isEmpty() const { return v_.empty(); }
  T &operator[](int i) { return v_[(size_t)i]; }
  const T &operator[](int i) const { return v_[(size_t)i]; }
  const T &at(int i) const { return v_[(size_t)i]; }
  typename std::vector<T>::iterator begin() { return v_.begin(); }
  typename std::vector<T>::iterator end() { return v_.end(); }
  typename std::vector<T>::const_iterator begin() const { return v_.begin(); }
  typename std::vector<T>::const_iterator end() const { return v_.end(); }
  void clear() { v_.clear(); }
 private:
  std::vector<T> v_;
};
class QThread {};
class QMutex { public: void lock() {} void unlock() {} };
class QThreadPool {
 public:
  static QThreadPool *globalInstance() { static QThreadPool p; return &p; }
  int maxThreadCount() const { return 1; }
};

namespace boost {
namespace fgo_detail {
struct placeholder1 {};
template <class R, class T, class A>
struct bound_mf1 {
  typedef R result_type;
  R (T::*f)(A);
  T *o;
  R operator()(A a) const { return (o->*f)(a); }
};
}  // namespace fgo_detail
template <class R, class T, class A>
fgo_detail::bound_mf1<R, T, A> bind(R (T::*f)(A), T *o, fgo_detail::placeholder1) {
  fgo_detail::bound_mf1<R, T, A> b; b.f = f; b.o = o; return b;
}
}  // namespace boost
namespace { boost::fgo_detail::placeholder1 _1; }

namespace QtConcurrent {
template <class T, class F>
QList<typename F::result_type> blockingMapped(const QList<T> &seq, F fn) {
  QList<typename F::result_type> out;
  for (int i = 0; i < seq.size(); ++i) out.push_back(fn(seq[i]));
  return out;
}
}  // namespace QtConcurrent
