// make_eigen_golden.cpp -- the reference's OWN vendored Eigen (3.3.90, /root/reference/include/Eigen) run on the
// small-matrix steps of the hot path, to pin the oracle's hand restatements of them.
//
// TEST INFRASTRUCTURE, build container only:
//     g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared -I/root/reference/include \
//         tests/golden/make_eigen_golden.cpp -o oracle/_ref/libeigen_ref.so
// (make_eigen_golden.py does this).  Nothing of Eigen is copied into this repository and the library never travels
// in git (oracle/_ref/ is ignored); what is committed are its OUTPUTS, tests/golden/eigen_golden.npz.
//
// Every function is the expression the reference (or PCL <= 1.10 on the reference's behalf, [PCL-recall]) evaluates,
// written with Eigen types so that Eigen's own code paths run:
//   eig_svd6_solve   delta_p = JacobiSVD<Matrix<double,6,6>>(hessian, ComputeFullU | ComputeFullV).solve(-score_gradient)
//                    (NDT::computeTransformation; Eigen: src/SVD/JacobiSVD.h:488,664)
//   eig_leaf         VoxelGridCovariance::applyFilter's per-leaf block: mean, single-pass covariance, SelfAdjointEigenSolver
//                    <Matrix3d>::compute, eigenvalue floor, cov = V D V^-1, icov = cov.inverse()
//                    (Eigen: src/Eigenvalues/SelfAdjointEigenSolver.h:405-553, Tridiagonalization.h:464-504, LU/InverseImpl.h:140-200)
//   eig_inv3         Matrix3d::inverse()  (src/PoseEstimator.cpp:64, src/PoseFuser.cpp; Eigen: LU/InverseImpl.h:140-200)
//   eig_init_guess   (Translation3f(tx,ty,0) * AngleAxisf(yaw, UnitZ)).matrix()  (src/PoseEstimator.cpp:22-24) and the
//                    prologue of computeTransformation: Affine3f.rotation().eulerAngles(0,1,2)  (Eigen: Geometry/EulerAngles.h:35-110,
//                    Geometry/Transform.h:1088-1121 -- rotation() of an Affine transform goes through a float JacobiSVD)
//   eig_step_matrix  the line search's final_transformation_ = (Translation<float,3>(x,y,z) * AngleAxis<float>(roll, UnitX) *
//                    AngleAxis<float>(pitch, UnitY) * AngleAxis<float>(yaw, UnitZ)).matrix()  (NDT::computeStepLengthMT)
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <cstring>
#include <limits>

extern "C" {

int eig_version(int v[3]) { v[0] = EIGEN_WORLD_VERSION; v[1] = EIGEN_MAJOR_VERSION; v[2] = EIGEN_MINOR_VERSION; return 0; }

// H3 row-major 3x3 over (tx, ty, yaw); g3 likewise.  The 6x6 / 6-vector PCL holds have these in rows/cols {0,1,5}
// and exact zeros elsewhere (SURVEY.md 8a note).  dp6 = the full solution, dp3 = its {0,1,5} entries.
void eig_svd6_solve(const double *H3, const double *g3, double *dp3, double *dp6, double *sv6) {
  Eigen::Matrix<double, 6, 6> hessian = Eigen::Matrix<double, 6, 6>::Zero();
  Eigen::Matrix<double, 6, 1> score_gradient = Eigen::Matrix<double, 6, 1>::Zero(), delta_p;
  const int ix[3] = {0, 1, 5};
  for (int i = 0; i < 3; ++i) {
    score_gradient(ix[i]) = g3[i];
    for (int j = 0; j < 3; ++j) hessian(ix[i], ix[j]) = H3[3 * i + j];
  }
  Eigen::JacobiSVD<Eigen::Matrix<double, 6, 6> > sv(hessian, Eigen::ComputeFullU | Eigen::ComputeFullV);
  delta_p = sv.solve(-score_gradient);
  for (int i = 0; i < 6; ++i) { dp6[i] = delta_p(i); sv6[i] = sv.singularValues()(i); }
  for (int i = 0; i < 3; ++i) dp3[i] = delta_p(ix[i]);
}

// One leaf of VoxelGridCovariance::applyFilter.  pt_sum[3] = sum of the points (fp64), cov_acc[9] row-major = the
// accumulated pt * pt^T (plus the identity for PCL <= 1.10, whose Leaf() starts cov_ at Identity).
// unbiased = 0: PCL <= 1.10 ((cov - 2 pt_sum mean^T)/n + mean mean^T, then * (n-1)/n); 1: PCL >= 1.11 ((cov - pt_sum mean^T)/(n-1)).
// Returns nr_points as PCL leaves it (-1: leaf rejected).
int eig_leaf(int n, const double *pt_sum_in, const double *cov_acc, double min_covar_eigvalue_mult, int unbiased,
             double *mean_out, double *cov_out, double *evals_out, double *evecs_out, double *icov_out) {
  Eigen::Vector3d pt_sum(pt_sum_in[0], pt_sum_in[1], pt_sum_in[2]);
  Eigen::Matrix3d cov_;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov_(i, j) = cov_acc[3 * i + j];
  int nr_points = n;
  Eigen::Vector3d mean_ = pt_sum;       // PCL: leaf.mean_ accumulated the points, pt_sum = leaf.mean_ before the division
  Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> eigensolver;
  Eigen::Matrix3d eigen_val, evecs_ = Eigen::Matrix3d::Zero(), icov_ = Eigen::Matrix3d::Zero();
  Eigen::Vector3d evals_ = Eigen::Vector3d::Zero();
  double min_covar_eigvalue;

  mean_ /= nr_points;
  if (!unbiased) {
    cov_ = (cov_ - 2 * (pt_sum * mean_.transpose())) / nr_points + mean_ * mean_.transpose();
    cov_ *= (nr_points - 1.0) / nr_points;
  } else {
    cov_ = (cov_ - pt_sum * mean_.transpose()) / (nr_points - 1.0);
  }
  eigensolver.compute(cov_);
  eigen_val = eigensolver.eigenvalues().asDiagonal();
  evecs_ = eigensolver.eigenvectors();
  bool rejected = false;
  if (eigen_val(0, 0) < 0 || eigen_val(1, 1) < 0 || eigen_val(2, 2) <= 0) {
    nr_points = -1; rejected = true;
  }
  if (!rejected) {
    min_covar_eigvalue = min_covar_eigvalue_mult * eigen_val(2, 2);
    if (eigen_val(0, 0) < min_covar_eigvalue) {
      eigen_val(0, 0) = min_covar_eigvalue;
      if (eigen_val(1, 1) < min_covar_eigvalue) eigen_val(1, 1) = min_covar_eigvalue;
      cov_ = evecs_ * eigen_val * evecs_.inverse();
    }
    evals_ = eigen_val.diagonal();
    icov_ = cov_.inverse();
    if (icov_.maxCoeff() == std::numeric_limits<float>::infinity() ||
        icov_.minCoeff() == -std::numeric_limits<float>::infinity())
      nr_points = -1;
  }
  for (int i = 0; i < 3; ++i) {
    mean_out[i] = mean_(i); evals_out[i] = rejected ? eigensolver.eigenvalues()(i) : evals_(i);
    for (int j = 0; j < 3; ++j) { cov_out[3 * i + j] = cov_(i, j); evecs_out[3 * i + j] = evecs_(i, j); icov_out[3 * i + j] = icov_(i, j); }
  }
  return nr_points;
}

void eig_inv3(const double *A, double *out) {
  Eigen::Matrix3d M;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) = A[3 * i + j];
  Eigen::Matrix3d R = M.inverse();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[3 * i + j] = R(i, j);
}

// src/PoseEstimator.cpp:22-24 and the prologue of NDT::computeTransformation.  M16 row-major 4x4 (float);
// trans3 = eig_transformation.translation(); euler_rot3 = eig_transformation.rotation().eulerAngles(0,1,2) (what PCL
// calls); euler_lin3 = eig_transformation.linear().eulerAngles(0,1,2) (the same without rotation()'s SVD round trip);
// rot9 = rotation() row-major.
void eig_init_guess(float tx, float ty, float yaw, float *M16, float *trans3, float *euler_rot3, float *euler_lin3, float *rot9) {
  Eigen::Translation3f init_translation(tx, ty, 0);
  Eigen::AngleAxisf init_rotation(yaw, Eigen::Vector3f::UnitZ());
  Eigen::Matrix4f init_guess = (init_translation * init_rotation).matrix();
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M16[4 * i + j] = init_guess(i, j);
  Eigen::Transform<float, 3, Eigen::Affine, Eigen::ColMajor> eig_transformation;
  eig_transformation.matrix() = init_guess;
  Eigen::Vector3f t = eig_transformation.translation();
  Eigen::Matrix3f R = eig_transformation.rotation();
  Eigen::Vector3f e = R.eulerAngles(0, 1, 2);
  Eigen::Matrix3f L = eig_transformation.linear();
  Eigen::Vector3f el = L.eulerAngles(0, 1, 2);
  for (int i = 0; i < 3; ++i) { trans3[i] = t(i); euler_rot3[i] = e(i); euler_lin3[i] = el(i); for (int j = 0; j < 3; ++j) rot9[3 * i + j] = R(i, j); }
}

// NDT::computeStepLengthMT: the float matrix of a trial, x_t = (x, y, z, roll, pitch, yaw) in fp64.
void eig_step_matrix(const double *x_t, float *M16) {
  Eigen::Matrix4f final_transformation_ =
      (Eigen::Translation<float, 3>(static_cast<float>(x_t[0]), static_cast<float>(x_t[1]), static_cast<float>(x_t[2])) *
       Eigen::AngleAxis<float>(static_cast<float>(x_t[3]), Eigen::Vector3f::UnitX()) *
       Eigen::AngleAxis<float>(static_cast<float>(x_t[4]), Eigen::Vector3f::UnitY()) *
       Eigen::AngleAxis<float>(static_cast<float>(x_t[5]), Eigen::Vector3f::UnitZ())).matrix();
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M16[4 * i + j] = final_transformation_(i, j);
}

}  // extern "C"
