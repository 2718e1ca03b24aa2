'use strict';
/* BWTC.js — drop-in front for the reference's `BWTC` object (J/BWTC_joined_.js:1696-1698,1827). */
var common = require('./common.js');

function rethrow(e) {
  if (e && typeof e.cjsCode === 'number') {
    if (e.cjsCode === -21) { throw new Error('Bad magic'); }                      // J/BWTC_joined_.js:559-565
    var g = new Error(e.message); g.errorCode = e.cjsCode; throw g;
  }
  throw e;
}

var BWTC = Object.create(null);
BWTC.MAGIC = 'bwtc';
BWTC.compressFile = function (inStream, outStream, props) {
  var level = 9;
  if (typeof props === 'number' && props >= 1 && props <= 9) { level = props; }   // J/BWTC_joined_.js:1702-1705
  var input = common.coerceInput(inStream);
  var result;
  // a stream input without .size makes the reference write varint(0) as the size field (J/BWTC_joined_.js:529-543, SURVEY W1)
  try { result = common.addon().bwtcCompress(input.bytes, level, input.hasSize ? 0 : 1 /* CJS_FLAG_SIZE_UNKNOWN */); } catch (e) { rethrow(e); }
  return common.deliver(result, outStream);
};
BWTC.decompressFile = function (inStream, outStream) {
  var input = common.coerceInput(inStream);
  var result;
  try { result = common.addon().bwtcDecompress(input.bytes); } catch (e) { rethrow(e); }
  return common.deliver(result, outStream);
};
module.exports = BWTC;
