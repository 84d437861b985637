function [eb, results] = sbtv_sapg(kind, y, op, c)
% [eb, results] = sbtv_sapg(kind, y, op, c)  - common body of the SAPG_algorithm_* shims.
%
% kind: 0 gaussian (w1,w2), 1 moffat (alpha,beta), 2 laplace (b).  op / c are the reference's structs
% (run_Gaussian_demo.m:34-39,186-204 and twins).  The likelihood closures op.f, op.gradF, op.grad_*,
% op.proxG, op.logPi are NOT called: the library evaluates the demos' stock expressions spectrally on the GPU
% from y, the PSF family and the current parameters (SURVEY.md §8 a-5).  The shim cannot verify that the
% handles in `op` ARE the stock closures, so it says so once per session (warning sbtv:closuresIgnored).
% MATLAB's randn stream is replaced by the device Philox generator seeded with op.seed (default 1).
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
persistent ctx warned
if isempty(ctx), ctx = sbtv_load(0); end
if isempty(warned) && any(isfield(op, {'gradF','proxG','logPi','f','g'}))
    warning('sbtv:closuresIgnored', ['op.gradF / op.proxG / op.logPi (and op.f, op.g, op.grad_*) are not called: the GPU path ' ...
            'evaluates the stock closures of run_*_demo.m (Gaussian likelihood, TV prior with chambolleit = %d). ' ...
            'A changed likelihood or prior is NOT picked up.'], getf(op,'chambolleit',25));
    warned = true;
end
names = {{'w1','w2'}, {'alpha','beta'}, {'b'}};
nm = names{kind+1};
[M, N] = size(y);
o = libstruct('sbtv_sapg_opts');
o.kind = kind; o.psf_size = getf(op,'psf_size',7);
o.samples = op.samples; o.warmup = getf(op,'warmup',100); o.burnIn = op.burnIn;
o.chambolleit = getf(op,'chambolleit',25);
o.share_gradients = 0; o.fix_sigma = getf(op,'fix_sigma',0);
o.lambda = getf(c,'lam',1) * op.lambda;  o.gamma = getf(c,'gam',1) * op.gamma;       % SAPG_algorithm_Guassian.m:30-31
o.th_init = op.th_init; o.min_th = op.min_th; o.max_th = op.max_th;
p_init = [0 0]; p_min = [0 0]; p_max = [0 0]; p_true = [0 0]; fix_p = int32([0 1]); c_p = [0 0];
for q = 1:numel(nm)
    p_init(q) = op.([nm{q} '_init']); p_min(q) = op.(['min_' nm{q}]); p_max(q) = op.(['max_' nm{q}]);
    t = op.(nm{q}); p_true(q) = t(1); fix_p(q) = int32(getf(op, ['fix_' nm{q}], 0)); c_p(q) = c.(nm{q});
end
o.p_init = p_init; o.p_min = p_min; o.p_max = p_max; o.p_true = p_true; o.fix_p = fix_p; o.c_p = c_p;
o.phi = getf(op,'phi',0);
o.sigma2_true = op.sigma^2; o.sigma2_init = op.sigma_init; o.sigma2_min = op.sigma_min; o.sigma2_max = op.sigma_max;
o.d_scale = op.d_scale; o.d_exp = op.d_exp; o.c_theta = c.theta; o.c_sigma = c.sigma;
o.seed = uint64(getf(op,'seed',1)); o.chain_offset = int32(getf(op,'chain_offset',0));
o.iter_offset = int32(getf(op,'iter_offset',0));
S = double(o.samples); W = max(double(o.warmup),1);
pth = libpointer('doublePtr', zeros(1,S)); psg = libpointer('doublePtr', zeros(1,S));
pps = libpointer('doublePtr', zeros(S,2)); plp = libpointer('doublePtr', zeros(1,S));
pwu = libpointer('doublePtr', zeros(1,W)); pgx = libpointer('doublePtr', zeros(1,S));
pgr = libpointer('doublePtr', zeros(S,4)); peb = libpointer('doublePtr', zeros(1,4));
pxl = libpointer('doublePtr', zeros(M,N));
x0 = []; if isfield(op,'X0'), x0 = op.X0; end
tic;
rc = calllib('libsbtv', 'sbtv_SAPG_algorithm', ctx, y, int32(M), int32(N), int32(1), o, x0, [], ...
             pth, pps, psg, plp, pwu, pgx, pgr, peb, pxl, [], [], int32(0));
if rc ~= 0, error('sbtv:SAPG', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
eb = peb.Value;
results.execTimeFindParameters = toc;                       % fields of SAPG_algorithm_Guassian.m:250-306
results.last_samp = S;
results.logPiTraceX = plp.Value; results.gXTrace = pgx.Value; results.logPiTrace_WU = pwu.Value(1:double(o.warmup));
results.theta_EB = eb(1); results.thetas = pth.Value; results.last_theta = results.thetas(end);
[results.mean_thetas, results.tol_thetas] = running_mean_tol(results.thetas, double(o.burnIn));
ps = reshape(pps.Value, S, 2); gr = reshape(pgr.Value, S, 4);
for q = 1:numel(nm)
    results.([nm{q} '_EB']) = eb(1+q); results.([nm{q} 's']) = ps(:,q)'; results.(['last_' nm{q}]) = ps(end,q);
    results.(['grad_' nm{q}]) = gr(:,1+q)'; results.(['c_' nm{q}]) = c.(nm{q});
    [results.(['mean_' nm{q} 's']), results.(['tol_' nm{q} 's'])] = running_mean_tol(ps(:,q)', double(o.burnIn));
end
results.sigma_EB = eb(4); results.sigmas = psg.Value; results.last_sigma = results.sigmas(end);
[results.mean_sigmas, results.tol_sigma] = running_mean_tol(results.sigmas, double(o.burnIn));
results.grad_theta = gr(:,1)'; results.grad_sigma = gr(:,4)';
results.c_theta = c.theta; results.c_sigma = c.sigma;
perr = libpointer('doublePtr', zeros(1,S));                 % results.err_psf (l2 with the matrix 2-norm, utils/l2.m)
rc = calllib('libsbtv', 'sbtv_err_psf', int32(kind), int32(o.psf_size), [ps(:,1); ps(:,2)], int32(S), p_true, o.phi, perr);
if rc ~= 0, error('sbtv:SAPG', '%s', calllib('libsbtv', 'sbtv_last_error', [])); end
results.err_psf = perr.Value;
results.Xlast_sample = reshape(pxl.Value, M, N);
results.options = op;
end

function [m, tol] = running_mean_tol(tr, burnIn)
% the running logs of SAPG_algorithm_Guassian.m:217-247: tol(ii) = |mean(tr(burnIn:ii)) - mean(tr(burnIn:ii-1))| /
% mean(tr(burnIn:ii-1)) for ii >= 2 (NaN while the window burnIn:ii-1 is empty, 0 at ii = 1), and
% m(ii - burnIn) = mean(tr(burnIn:ii)) for ii > burnIn
S = numel(tr);
tol = zeros(1, S); m = zeros(1, max(S - burnIn, 0));
for ii = 2:S
    tol(ii) = abs(mean(tr(burnIn:ii)) - mean(tr(burnIn:ii-1))) / mean(tr(burnIn:ii-1));
    if ii > burnIn, m(ii - burnIn) = mean(tr(burnIn:ii)); end
end
end

function v = getf(s, name, default)
if isfield(s, name), v = s.(name); else, v = default; end
end
