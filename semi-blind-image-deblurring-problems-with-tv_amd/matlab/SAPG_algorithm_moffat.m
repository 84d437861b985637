function [theta_EB, alpha_EB, beta_EB, sigma2_EB, results] = SAPG_algorithm_moffat(y, op)
% Drop-in replacement of SAPG/SAPG_algorithm_moffat.m:7; the step scales are the ones hard-coded there (:135-138).
c = struct('theta', 0.1, 'alpha', 10, 'beta', 1e4, 'sigma', 1e4, 'lam', 1, 'gam', 1);
[eb, results] = sbtv_sapg(1, y, op, c);
theta_EB = eb(1); alpha_EB = eb(2); beta_EB = eb(3); sigma2_EB = eb(4);
end
