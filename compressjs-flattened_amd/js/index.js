'use strict';
/* index.js — the two algorithm objects of the hot path, as `require('compressjs')` would expose them
 * (NPM/main.js:2-28 lists every algorithm; only Bzip2 and BWTC are in scope here). */
module.exports = { Bzip2: require('./Bzip2.js'), BWTC: require('./BWTC.js'), native: function () { return require('./common.js').addon(); } };
