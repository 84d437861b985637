function [theta_EB, w1_EB, w2_EB, sigma_EB, results] = SAPG_algorithm_Guassian(y, op, c)
% Drop-in replacement of SAPG/SAPG_algorithm_Guassian.m:7 (same signature; the reference's results fields incl. the mean_* / tol_* logs; never executed: no MATLAB here) running the
% MYULA chain and the SAPG updates device-resident on the MI355X through libsbtv.so.  See sbtv_sapg.m.
[eb, results] = sbtv_sapg(0, y, op, c);
theta_EB = eb(1); w1_EB = eb(2); w2_EB = eb(3); sigma_EB = eb(4);
end
