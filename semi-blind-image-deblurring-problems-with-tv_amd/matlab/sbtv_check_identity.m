function sbtv_check_identity(who, P, PT, definedP, definedPT, M, N, nameP, namePT)
% sbtv_check_identity(who, P, PT, definedP, definedPT, M, N, nameP, namePT)
% The analysis operators 'P' / 'PT' of SALSA_v2 / csalsa ('P1','P1T','P2','P2T' of CoRAL) enter the reference's iteration
% (r = ATy + mu*P(u+bu), PTx = PT(x): SALSA/SALSA_v2.m:389,434,438; CSALSA_v2.m:402,473; CoRAL_v2.m:349-350,411,417-418).
% The GPU path has P = PT = identity only, so a pair that is given must BE the identity: it is probed with a fixed image and
% anything else is an error (ignoring it silently would return a different result than the reference).  The reference's own
% pairing check comes first (SALSA_v2.m:251-253, CSALSA_v2.m:264-266, CoRAL_v2.m:142-144,152-154).
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
if xor(definedP, definedPT)
    error(['If you give ' nameP ' you must also give ' namePT ', and vice versa.']);
end
if ~definedP, return; end
probe = reshape(mod((1:M*N) * 0.6180339887498949, 1), M, N);      % fixed, non-symmetric, non-constant
ops = {P, PT}; names = {nameP, namePT};
for k = 1:2
    f = ops{k};
    if ~isa(f, 'function_handle')
        error(who, '''%s'' must be a function handle', names{k});
    end
    out = f(probe);
    if ~isequal(size(out), [M N]) || max(abs(out(:) - probe(:))) > 1e-12
        error(who, '''%s'' is not the identity: the GPU path runs the TV problem with %s = %s = identity only', names{k}, nameP, namePT);
    end
end
end
