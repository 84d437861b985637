function [theta_EB, b_EB, sigma2_EB, results] = SAPG_algorithm_laplace(y, op)
% Drop-in replacement of SAPG/SAPG_algorithm_laplace.m:7; the step scales are the ones hard-coded there (:139-141).
c = struct('theta', 0.01, 'b', 100, 'sigma', 1e4, 'lam', 1, 'gam', 1);
[eb, results] = sbtv_sapg(2, y, op, c);
theta_EB = eb(1); b_EB = eb(2); sigma2_EB = eb(4);
end
